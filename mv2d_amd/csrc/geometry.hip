// Geometry / gather kernels of the MV2D hot path (gfx950, wave64).  HBM/latency-bound integer, fp64 and
// byte work: coalesced position-major rows, LDS bitmasks, wavefront ballots — no MFMA here on purpose.
// Compiled with -ffp-contract=off: the fp64/fp32 comparisons against box edges must follow the reference's
// operation order exactly (integer / boolean outputs are a bit-exact target).
#include "common.h"

namespace {

constexpr int C = 256;

// ------------------------------------------------------------------------------------------------
// a3/a5/a7: per-RoI camera (RH/mv2d_head.py:51-72), intrinsics feature (:95-101) and
//           inverse(K_roi @ E^T).float() of center2lidar (RH/utils/query_generator.py:333-341)
// ------------------------------------------------------------------------------------------------
__device__ bool inverse4x4(const double* m, double* inv) {
    // Gauss-Jordan with partial pivoting (same pivoting rule as LAPACK getrf)
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c; double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r) { double v = fabs(a[r][c]); if (v > best) { best = v; piv = r; } }
        if (best == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] = a[c][j] / d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = a[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
    return true;
}

__device__ __forceinline__ void box_params_row(int r, const float* __restrict__ rois, const double* __restrict__ viewK, const double* __restrict__ viewE,
                                               double* __restrict__ K_roi, float* __restrict__ intr, int ld_intr, float* __restrict__ minv,
                                               int R, float roi_size, float intr_scale, float min_size) {
    if (r >= R) return;
    const float* b = rois + r * 5;
    const int v = (int)b[0];
    const float w = b[3] - b[1], h = b[4] - b[2];
    const float sx = roi_size / w, sy = roi_size / h;
    double K[16];
    for (int i = 0; i < 16; ++i) K[i] = viewK[v * 16 + i];
    K[2] = (K[2] - (double)b[1]) - (double)(0.5f / sx);
    K[6] = (K[6] - (double)b[2]) - (double)(0.5f / sy);
    for (int j = 0; j < 4; ++j) { K[j] = K[j] * (double)sx; K[4 + j] = K[4 + j] * (double)sy; }
    const bool small = (w < min_size) || (h < min_size);
    for (int i = 0; i < 16; ++i) {
        if (K_roi) K_roi[r * 16 + i] = K[i];
        float f = (float)K[i] * intr_scale;
        f = small ? 0.f : f;
        intr[(long long)r * ld_intr + i] = fminf(fmaxf(f, -5e3f), 5e3f);     // clamp of query_generator.py:369
    }
    // lidar2img = K_roi @ E^T  (E = "extrinsics", stored transposed), sequential k accumulation like bmm
    const double* E = viewE + v * 16;
    double L[16], Li[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc = acc + K[i * 4 + k] * E[j * 4 + k];
            L[i * 4 + j] = acc;
        }
    bool ok = inverse4x4(L, Li);
    for (int i = 0; i < 16; ++i) minv[r * 16 + i] = ok ? (float)Li[i] : __builtin_nanf("");
}

__global__ void box_params_kernel(const float* __restrict__ rois, const double* __restrict__ viewK, const double* __restrict__ viewE,
                                  double* __restrict__ K_roi, float* __restrict__ intr, int ld_intr, float* __restrict__ minv,
                                  int R, float roi_size, float intr_scale, float min_size) {
    box_params_row(blockIdx.x * blockDim.x + threadIdx.x, rois, viewK, viewE, K_roi, intr, ld_intr, minv, R, roi_size, intr_scale, min_size);
}

// inverse(K_roi @ E^T).float() for arbitrary per-RoI fp64 matrices (module-level QueryGenerator.center2lidar)
__global__ void lidar2img_inverse_kernel(const double* __restrict__ K_roi, const double* __restrict__ E, float* __restrict__ minv, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double L[16], Li[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc = acc + K_roi[r * 16 + i * 4 + k] * E[r * 16 + j * 4 + k];
            L[i * 4 + j] = acc;
        }
    bool ok = inverse4x4(L, Li);
    for (int i = 0; i < 16; ++i) minv[r * 16 + i] = ok ? (float)Li[i] : __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------------
// a7/a8/a13: center2lidar mat-vec, pc_range normalisation (no clamp — RH/mv2d_t_head.py:51-57),
//            pos2posemb3d (MU/pe.py:21-33): one wave per RoI, lanes over the 384 sine channels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refpoint_posemb_kernel(const float* __restrict__ center_pred, int ld_cp, const float* __restrict__ minv,
                                                              const float* __restrict__ dim_t, float* __restrict__ xyz, float* __restrict__ ref,
                                                              float* __restrict__ posemb, int R, float pc0, float pc1, float pc2,
                                                              float pd0, float pd1, float pd2) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float u = center_pred[(long long)r * ld_cp + 0], v = center_pred[(long long)r * ld_cp + 1], d = center_pred[(long long)r * ld_cp + 2];
    const float c[4] = {u * d, v * d, d, 1.0f};
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = acc + minv[r * 16 + i * 4 + k] * c[k];
        p[i] = acc;
    }
    const float n0 = (p[0] - pc0) / pd0, n1 = (p[1] - pc1) / pd1, n2 = (p[2] - pc2) / pd2;
    if (lane < 3) {
        xyz[r * 3 + lane] = p[lane];
        ref[r * 3 + lane] = lane == 0 ? n0 : (lane == 1 ? n1 : n2);
    }
    const float two_pi = 6.283185307179586f;
    const float py = n1 * two_pi, px = n0 * two_pi, pz = n2 * two_pi;    // output order (y | x | z)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ch = lane + 64 * j;           // 0..383
        const int axis = ch >> 7, i = ch & 127;
        const float pos = axis == 0 ? py : (axis == 1 ? px : pz);
        const float a = pos / dim_t[i];
        posemb[(long long)r * 384 + ch] = (i & 1) ? cosf(a) : sinf(a);
    }
}

// pos2posemb3d alone (MU/pe.py:21-33) for the module-level CrossAttentionBoxHead.position_embedding
__global__ __launch_bounds__(256) void posemb3d_kernel(const float* __restrict__ ref, const float* __restrict__ dim_t, float* __restrict__ posemb, int R) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float two_pi = 6.283185307179586f;
    const float py = ref[r * 3 + 1] * two_pi, px = ref[r * 3 + 0] * two_pi, pz = ref[r * 3 + 2] * two_pi;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ch = lane + 64 * j;
        const int axis = ch >> 7, i = ch & 127;
        const float pos = axis == 0 ? py : (axis == 1 ? px : pz);
        const float a = pos / dim_t[i];
        posemb[(long long)r * 384 + ch] = (i & 1) ? cosf(a) : sinf(a);
    }
}

// ------------------------------------------------------------------------------------------------
// a4: RoIAlign (mmcv 1.6.1 semantics: aligned, avg, adaptive sampling grid) on position-major maps.
//     One block per (RoI, bin row); every bilinear tap is a fully coalesced C*4-byte row read.
//     maps: up to two [V*h*w, 256] fp32 maps (feature, PE) -> out key16 (fp16) [R, 49, 256] each (+ optional fp32)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ map0, const float* __restrict__ map1, const float* __restrict__ rois,
                                                        unsigned short* __restrict__ out0, unsigned short* __restrict__ out1,
                                                        float* __restrict__ out0_f32, float* __restrict__ out1_f32, int H, int W,
                                                        float spatial_scale, int sampling_ratio, const int* __restrict__ map1_index,
                                                        int out1_is_sum, int R, unsigned short* __restrict__ out0_lo,
                                                        unsigned short* __restrict__ out1_lo, unsigned char* __restrict__ out0_lo8,
                                                        unsigned char* __restrict__ out1_lo8, int* __restrict__ lo8_flag) {
    // one block per RoI and bin row; wave w takes the bins w, w + 4 of the row, lane l the channels 4l .. 4l+3: every
    // bilinear tap is one 16-byte load per lane (a full 1 KB row per wave), every output one 8-byte (key16) / 16-byte (fp32) store.
    // XCD-aware block map (block b runs on XCD b % 8): the 7 bin rows of a RoI tap overlapping map rows, so they take consecutive slots
    // of ONE XCD and share its L2 (a (R, 7) grid ran them R blocks apart).  Speed only; any map is correct.
    const int slot = blockIdx.x >> 3, ph = slot % 7, r = (slot / 7) * 8 + (blockIdx.x & 7);
    if (r >= R) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
    const float* b = rois + r * 5;
    const int v = (int)b[0];
    const float x1 = b[1] * spatial_scale - 0.5f, y1 = b[2] * spatial_scale - 0.5f;
    const float x2 = b[3] * spatial_scale - 0.5f, y2 = b[4] * spatial_scale - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / 7.0f, bh = rh / 7.0f;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / 7.0f);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / 7.0f);
    const float count = (float)max(gh * gw, 1);
    const long long vbase = (long long)v * H * W;
    const int nmaps = map1 ? 2 : 1;
    auto ld = [&](const float* m, long long q) { return *reinterpret_cast<const float4*>(m + q * C + c); };
    auto tap4 = [](float w1, const float4& a, float w2, const float4& bq, float w3, const float4& cq, float w4, const float4& d, float4& s) {
        s.x += w1 * a.x + w2 * bq.x + w3 * cq.x + w4 * d.x; s.y += w1 * a.y + w2 * bq.y + w3 * cq.y + w4 * d.y;
        s.z += w1 * a.z + w2 * bq.z + w3 * cq.z + w4 * d.z; s.w += w1 * a.w + w2 * bq.w + w3 * cq.w + w4 * d.w;
    };
    for (int pw = wave; pw < 7; pw += 4) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        for (int iy = 0; iy < gh; ++iy) {
            const float yy = y1 + ph * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float xx = x1 + pw * bw + (ix + 0.5f) * bw / gw;
                if (yy < -1.0f || yy > (float)H || xx < -1.0f || xx > (float)W) continue;
                float y = fmaxf(yy, 0.f), x = fmaxf(xx, 0.f);
                int yl = (int)y, xl = (int)x, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const long long q1 = vbase + (long long)yl * W + xl, q2 = vbase + (long long)yl * W + xh;
                const long long q3 = vbase + (long long)yh * W + xl, q4 = vbase + (long long)yh * W + xh;
                tap4(w1, ld(map0, q1), w2, ld(map0, q2), w3, ld(map0, q3), w4, ld(map0, q4), s0);
                if (nmaps == 2) {
                    long long p1 = q1, p2 = q2, p3 = q3, p4 = q4;
                    if (map1_index) {   // map1 rows are compacted: row = map1_index[position]; -1 rows only ever carry weight 0
                        p1 = max(map1_index[q1], 0); p2 = max(map1_index[q2], 0); p3 = max(map1_index[q3], 0); p4 = max(map1_index[q4], 0);
                    }
                    tap4(w1, ld(map1, p1), w2, ld(map1, p2), w3, ld(map1, p3), w4, ld(map1, p4), s1);
                }
            }
        }
        const long long o = ((long long)r * 49 + ph * 7 + pw) * C + c;
        s0 = make_float4(s0.x / count, s0.y / count, s0.z / count, s0.w / count);
        // key16 outputs (common.h: fp16 since round 4); out*_lo: the remainder x - key16(x) next to the value: the fp32-class hi + lo rows of the
        // index-exact route; out*_lo8 (round 6): the same remainder as 256-byte e4m3 rows (common.h "lo8": what the cross attention gathers)
        auto put = [&](unsigned short* hi, unsigned short* lo, unsigned char* lo8, const float4& t) {
            if (lo || lo8) {
                uint2 hh, ll;
                split_k16x2(t.x, t.y, hh.x, ll.x);
                split_k16x2(t.z, t.w, hh.y, ll.y);
                *reinterpret_cast<uint2*>(hi + o) = hh;
                if (lo) *reinterpret_cast<uint2*>(lo + o) = ll;
                if (lo8) *reinterpret_cast<unsigned int*>(lo8 + o) = lo8_pack4_flag(ll.x, ll.y, lo8_flag);
            } else {
                *reinterpret_cast<uint2*>(hi + o) = make_uint2(pack_k16x2(t.x, t.y), pack_k16x2(t.z, t.w));
            }
        };
        if (out0) put(out0, out0_lo, out0_lo8, s0);
        if (out0_f32) *reinterpret_cast<float4*>(out0_f32 + o) = s0;
        if (nmaps == 2) {
            s1 = make_float4(s1.x / count, s1.y / count, s1.z / count, s1.w / count);
            if (out1) {
                const float4 t = out1_is_sum ? make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w) : s1;
                put(out1, out1_lo, out1_lo8, t);
            }
            if (out1_f32) *reinterpret_cast<float4*>(out1_f32 + o) = s1;
        }
    }
}

// Backward of roi_align_kernel w.r.t. one map (training): the gradient of every output bin is spread over the bilinear taps
// of its samples with the forward's weights / count.  Same grid and loops as the forward; fp32 hardware atomics into the (zeroed) map
// gradient, so the summation order — not the values' set — varies from run to run (mmcv's RoIAlign backward does the same).
// index: optional position -> row of a compacted map (rows < 0 are skipped).
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gmap,
                                                            const int* __restrict__ index, int H, int W, float spatial_scale, int sampling_ratio) {
    const int r = blockIdx.x, ph = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = 4 * lane;
    const float* b = rois + r * 5;
    const int v = (int)b[0];
    const float x1 = b[1] * spatial_scale - 0.5f, y1 = b[2] * spatial_scale - 0.5f;
    const float x2 = b[3] * spatial_scale - 0.5f, y2 = b[4] * spatial_scale - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / 7.0f, bh = rh / 7.0f;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / 7.0f);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / 7.0f);
    const float count = (float)max(gh * gw, 1);
    const long long vbase = (long long)v * H * W;
    auto add = [&](long long q, float wgt, const float4& g) {
        if (index) {
            const int row = index[q];
            if (row < 0) return;
            q = row;
        }
        float* d = gmap + q * C + c;
        unsafeAtomicAdd(d, wgt * g.x); unsafeAtomicAdd(d + 1, wgt * g.y); unsafeAtomicAdd(d + 2, wgt * g.z); unsafeAtomicAdd(d + 3, wgt * g.w);
    };
    for (int pw = wave; pw < 7; pw += 4) {
        float4 g = *reinterpret_cast<const float4*>(gout + ((long long)r * 49 + ph * 7 + pw) * C + c);
        g = make_float4(g.x / count, g.y / count, g.z / count, g.w / count);
        for (int iy = 0; iy < gh; ++iy) {
            const float yy = y1 + ph * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float xx = x1 + pw * bw + (ix + 0.5f) * bw / gw;
                if (yy < -1.0f || yy > (float)H || xx < -1.0f || xx > (float)W) continue;
                float y = fmaxf(yy, 0.f), x = fmaxf(xx, 0.f);
                int yl = (int)y, xl = (int)x, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
                add(vbase + (long long)yl * W + xl, hy * hx, g);
                add(vbase + (long long)yl * W + xh, hy * lx, g);
                add(vbase + (long long)yh * W + xl, ly * hx, g);
                add(vbase + (long long)yh * W + xh, ly * lx, g);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a9: epipolar box correlation (RH/utils/box_correlation.py:196-398, 'topk_matched:k:thr:ratio').
//     One block per (RoI r, destination view b of r's sample).  fp64 projection, fp32 compares, integer outputs.
//     match[r][b][rank] = global RoI id or -1.  V = views per sample; a batch of samples numbers its views consecutively
//     (view a belongs to sample a / V), correlation stays inside a sample, trans is [all views][V][16].
// ------------------------------------------------------------------------------------------------
constexpr int MAX_PER_VIEW = 1024;

struct BoxCorrArgs {
    const float* rois; const int* view_start; const double* trans; const float* lin; const float* depths; int* match;
    int V, ss, D, topk; float img_w_m1, img_h_m1, depth_start, iou_thr, ratio;
};

__device__ __forceinline__ void box_corr_block(int r, int bl, const BoxCorrArgs& A) {
    const float* __restrict__ rois = A.rois; const int* __restrict__ view_start = A.view_start; const double* __restrict__ trans = A.trans;
    const float* __restrict__ lin = A.lin; const float* __restrict__ depths = A.depths; int* __restrict__ match = A.match;
    const int V = A.V, ss = A.ss, D = A.D, topk = A.topk;
    const float img_w_m1 = A.img_w_m1, img_h_m1 = A.img_h_m1, depth_start = A.depth_start, iou_thr = A.iou_thr, ratio = A.ratio;
    __shared__ float su[128], sv[128];
    __shared__ int svalid[128];
    __shared__ float siou[MAX_PER_VIEW];
    __shared__ float red[4][2];
    __shared__ int flag[2];
    const int tid = threadIdx.x;
    int* out = match + ((long long)r * V + bl) * topk;
    for (int i = tid; i < topk; i += 128) out[i] = -1;
    const float* rb = rois + r * 5;
    const int a = (int)rb[0];
    const int b = (a / V) * V + bl;                        // destination view (global index)
    const int start = view_start[b], nb = view_start[b + 1] - start;
    if (a == b || nb == 0) return;
    if (tid < 2) flag[tid] = 0;
    __syncthreads();
    // ---- project the ss*ss*D samples of RoI r into view b
    const int ns = ss * ss * D;   // <= 128 (host-checked)
    bool valid = false;
    float u32 = 0.f, v32 = 0.f;
    if (tid < ns) {
        const int pt = tid / D, dk = tid - pt * D;
        const int iy = pt / ss, ix = pt - iy * ss;
        const float wbox = rb[3] - rb[1], hbox = rb[4] - rb[2];
        const float x = rb[1] + wbox * lin[ix], y = rb[2] + hbox * lin[iy];
        const double d = (double)depths[dk];
        const double hx = (double)x * d, hy = (double)y * d;
        const double* T = trans + ((long long)a * V + bl) * 16;
        double cam[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
            acc = acc + T[i * 4 + 0] * hx;
            acc = acc + T[i * 4 + 1] * hy;
            acc = acc + T[i * 4 + 2] * d;
            acc = acc + T[i * 4 + 3] * 1.0;
            cam[i] = acc;
        }
        const double zc = cam[2] < 1e-2 ? 1e-2 : cam[2];      // clamp_min(1e-2); NaN stays NaN like torch
        const double u = cam[0] / zc, vv = cam[1] / zc;
        valid = !(cam[2] < (double)depth_start) && (0.0 <= u) && (u <= (double)img_w_m1) && (0.0 <= vv) && (vv <= (double)img_h_m1);
        u32 = (float)u; v32 = (float)vv;
    }
    su[tid] = u32; sv[tid] = v32; svalid[tid] = valid ? 1 : 0;
    if (valid) flag[0] = 1;
    __syncthreads();
    if (!flag[0]) return;
    // ---- does any valid sample fall inside any RoI of view b? (closed intervals, :299-300)
    if (valid) {
        bool hit = false;
        for (int m = 0; m < nb && !hit; ++m) {
            const float* mb = rois + (long long)(start + m) * 5;
            hit = (mb[1] <= u32) && (u32 <= mb[3]) && (mb[2] <= v32) && (v32 <= mb[4]);
        }
        if (hit) flag[1] = 1;
    }
    __syncthreads();
    if (!flag[1]) return;
    // ---- bounding rectangle of ALL valid projected samples (:347-355), sentinels +-1e4
    float umin = valid ? u32 : 1e4f, vmin = valid ? v32 : 1e4f, umax = valid ? u32 : -1e4f, vmax = valid ? v32 : -1e4f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        umin = fminf(umin, __shfl_xor(umin, o, 64)); vmin = fminf(vmin, __shfl_xor(vmin, o, 64));
        umax = fmaxf(umax, __shfl_xor(umax, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = umin; red[1][tid >> 6] = vmin; red[2][tid >> 6] = umax; red[3][tid >> 6] = vmax; }
    __syncthreads();
    const float a0 = fminf(red[0][0], red[0][1]), a1 = fminf(red[1][0], red[1][1]);
    const float a2 = fmaxf(red[2][0], red[2][1]), a3 = fmaxf(red[3][0], red[3][1]);
    // ---- IoU against every RoI of view b (:385-398)
    for (int m = tid; m < nb; m += 128) {
        const float* mb = rois + (long long)(start + m) * 5;
        const float xs = fmaxf(a0, mb[1]), ys = fmaxf(a1, mb[2]), xe = fminf(a2, mb[3]), ye = fminf(a3, mb[4]);
        const float w = fmaxf(xe - xs, 0.f), h = fmaxf(ye - ys, 0.f);
        const float inter = w * h;
        const float area_a = (a2 - a0) * (a3 - a1), area_b = (mb[3] - mb[1]) * (mb[4] - mb[2]);
        const float uni = area_a + area_b - inter;
        siou[m] = inter / (uni + 1e-4f);
    }
    __syncthreads();
    // ---- top-k by IoU, descending, stable (lower index first among equals); keep rule of :374
    float top = 0.f;
    for (int j = 0; j < nb; ++j) top = fmaxf(top, siou[j]);
    for (int m = tid; m < nb; m += 128) {
        const float x = siou[m];
        int rank = 0;
        for (int j = 0; j < nb; ++j) { const float y = siou[j]; rank += (y > x || (y == x && j < m)) ? 1 : 0; }
        if (rank < topk) {
            const bool keep = ((x > ratio * top) || (x > iou_thr)) && (x > 0.f);
            out[rank] = keep ? (start + m) : -1;
        }
    }
}

__global__ __launch_bounds__(128) void box_corr_kernel(BoxCorrArgs A) { box_corr_block(blockIdx.x, blockIdx.y, A); }

// The feature-independent geometry of a frame in ONE launch (round 4: a one-sample frame is bound by its NUMBER of kernels, 4.6 us per
// dispatch): blocks (r < R, b) = box correlation of RoI r with view b; then, in row y = 0, ceil(R / 128) blocks of per-RoI cameras
// (box_params_row) and the blocks that clear the per-frame flag / mask bytes (the engine's zbuf).  The three parts are independent.
struct FrameGeoArgs {
    BoxCorrArgs corr;
    const double* viewK; const double* viewE; double* K_roi; float* intr; int ld_intr; float* minv; float roi_size, intr_scale, min_size;
    uint4* zero_ptr; long long zero_vec16;      // 16-byte words to clear
    int R;
};
constexpr int GEO_ZERO_PER_BLOCK = 128 * 8;    // uint4 per block

__global__ __launch_bounds__(128) void frame_geometry_kernel(FrameGeoArgs G) {
    const int bx = blockIdx.x;
    if (bx < G.R) { box_corr_block(bx, blockIdx.y, G.corr); return; }
    if (blockIdx.y != 0) return;
    const int np = (G.R + 127) / 128;
    if (bx < G.R + np) {
        box_params_row((bx - G.R) * 128 + threadIdx.x, G.corr.rois, G.viewK, G.viewE, G.K_roi, G.intr, G.ld_intr, G.minv, G.R, G.roi_size, G.intr_scale,
                       G.min_size);
        return;
    }
    const long long z0 = (long long)(bx - G.R - np) * GEO_ZERO_PER_BLOCK;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long i = z0 + k * 128 + threadIdx.x;
        if (i < G.zero_vec16) G.zero_ptr[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ------------------------------------------------------------------------------------------------
// a11/a12: T-path masks -> compacted key list + CSR (RH/utils/box_correlation.py:101-115,147-157 and
//          RH/mv2d_t_head.py:67-88).  cell centre c = (i + 0.5) * stride - 0.5; in-RoI iff
//          c + 0.5*stride + e*stride >= x1  and  c - 0.5*stride - e*stride <= x2  (both axes).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void roi_cell_range(const float* rb, int h, int w, float stride, float expand, int& y0, int& y1, int& x0, int& x1) {
    // explicit per-cell test (fp32, reference op order); the ranges are small so a scan of the axis is cheap
    x0 = w; x1 = -1; y0 = h; y1 = -1;
    for (int i = 0; i < w; ++i) {
        const float c = ((float)i + 0.5f) * stride - 0.5f;
        const bool in = ((c + 0.5f * stride) + expand * stride >= rb[1]) && ((c - 0.5f * stride) - expand * stride <= rb[3]);
        if (in) { x0 = min(x0, i); x1 = max(x1, i); }
    }
    for (int i = 0; i < h; ++i) {
        const float c = ((float)i + 0.5f) * stride - 0.5f;
        const bool in = ((c + 0.5f * stride) + expand * stride >= rb[2]) && ((c - 0.5f * stride) - expand * stride <= rb[4]);
        if (in) { y0 = min(y0, i); y1 = max(y1, i); }
    }
}

// S path (round 5): the cells the bilinear taps of RoIAlign can touch, EXACTLY -- the first and last sample coordinate of an axis evaluated with
// roi_align_kernel's own fp32 expressions (aligned = True, adaptive grid, 7 bins; sample (bin, i) is monotonic in both), tap = (int) max(x, 0)
// and its right / lower neighbour, clamped like the kernel clamps them.  The list the PE block is evaluated on shrinks by ~1 cell per axis and
// RoI against the "RoI expanded by one cell" rectangle it replaces (141 k -> ~110 k positions per 16 cfg2_s samples); every tap, also a
// zero-weight one, is still listed, so the RoI-aligned rows are bitwise the same.
__device__ __forceinline__ void roi_tap_range(const float* rb, int h, int w, float spatial_scale, int& y0, int& y1, int& x0, int& x1) {
    auto axis = [&](float lo_px, float hi_px, int n, int& c0, int& c1) {
        const float a = lo_px * spatial_scale - 0.5f, b = hi_px * spatial_scale - 0.5f;
        const float len = b - a, bin = len / 7.0f;
        const int g = (int)ceilf(len / 7.0f);
        if (g <= 0) { c0 = n; c1 = -1; return; }                           // (degenerate RoI: no sample point, count = 1, zero output)
        const float first = a + 0 * bin + (0 + 0.5f) * bin / g, last = a + 6 * bin + ((g - 1) + 0.5f) * bin / g;
        if (last < -1.0f || first > (float)n) { c0 = n; c1 = -1; return; }    // every sample of the axis is skipped
        const int lo = min((int)fmaxf(first, 0.f), n - 1), hi = min((int)fmaxf(last, 0.f), n - 1);
        c0 = lo;
        c1 = min(hi + 1, n - 1);
    };
    axis(rb[1], rb[3], w, x0, x1);
    axis(rb[2], rb[4], h, y0, y1);
    if (x1 < x0 || y1 < y0) { x0 = w; x1 = -1; y0 = h; y1 = -1; }
}

// rect[r] = (view, y0, y1, x0, x1) and roi_mask[P] |= own-view rect   (roi_mask pre-zeroed)
__global__ __launch_bounds__(64) void csr_mark_kernel(const float* __restrict__ rois, int* __restrict__ rect, unsigned char* __restrict__ roi_mask,
                                                      int h, int w, float stride, float expand) {
    const int r = blockIdx.x, lane = threadIdx.x;
    __shared__ int sr[4];
    const float* rb = rois + r * 5;
    if (lane == 0) {
        int y0, y1, x0, x1;
        if (expand < 0.f) roi_tap_range(rb, h, w, 1.0f / stride, y0, y1, x0, x1);      // S path: the RoIAlign tap cells
        else roi_cell_range(rb, h, w, stride, expand, y0, y1, x0, x1);
        sr[0] = y0; sr[1] = y1; sr[2] = x0; sr[3] = x1;
        rect[r * 5 + 0] = (int)rb[0]; rect[r * 5 + 1] = y0; rect[r * 5 + 2] = y1; rect[r * 5 + 3] = x0; rect[r * 5 + 4] = x1;
    }
    __syncthreads();
    const int v = (int)rb[0], y0 = sr[0], y1 = sr[1], x0 = sr[2], x1 = sr[3];
    if (y1 < y0 || x1 < x0) return;
    const int nw = x1 - x0 + 1, n = (y1 - y0 + 1) * nw;
    for (int i = lane; i < n; i += 64) {
        const int y = y0 + i / nw, x = x0 + i % nw;
        roi_mask[((long long)v * h + y) * w + x] = 1;
    }
}

// pos2s[P] = index into the compacted key list or -1 (not in any RoI rect, or padding),
// s2pos[S] = flat (v,y,x) position, *S = count.  Row-major (v,y,x) order like the reference's boolean indexing.
// One block per segment of SCAN_SEG cells: a block first counts the kept cells of all earlier segments itself (a few KB of mask
// bytes per thread-block, cheaper than a second launch or a look-back chain), then scans its own segment.  (A batch of 4 samples
// is P = 67584 cells: the single-block version of round 1 took 40 us there.)
constexpr int SCAN_SEG = 16384, SCAN_CPT = 16;      // cells per block and per thread

__device__ __forceinline__ unsigned int kept16(const unsigned char* __restrict__ roi_mask, const unsigned char* __restrict__ pad_mask, int j0, int P,
                                               bool vec_ok) {
    unsigned int bits = 0u;
    if (vec_ok && j0 + 15 < P) {
        const uint4 a = *reinterpret_cast<const uint4*>(roi_mask + j0);
        const uint4 b = *reinterpret_cast<const uint4*>(pad_mask + j0);
        const unsigned int aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned int ra = (aw[k >> 2] >> ((k & 3) * 8)) & 0xffu, pb = (bw[k >> 2] >> ((k & 3) * 8)) & 0xffu;
            bits |= ((ra != 0u && pb == 0u) ? 1u : 0u) << k;
        }
    } else {
        for (int k = 0; k < 16; ++k) { const int i = j0 + k; if (i < P && roi_mask[i] && !pad_mask[i]) bits |= 1u << k; }
    }
    return bits;
}

__device__ __forceinline__ void csr_scan_positions_block(int blk, int nblk, const unsigned char* __restrict__ roi_mask, const unsigned char* __restrict__ pad_mask,
                                                         int* __restrict__ pos2s, int* __restrict__ s2pos, int* __restrict__ S_out, int P) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg0 = blk * SCAN_SEG;
    const bool vec_ok = (P & 15) == 0 && ((uintptr_t)roi_mask & 15) == 0 && ((uintptr_t)pad_mask & 15) == 0 && ((uintptr_t)pos2s & 15) == 0;
    // ---- kept cells before this segment
    int before = 0;
    for (int j0 = tid * 16; j0 < seg0; j0 += 1024 * 16) before += __popc(kept16(roi_mask, pad_mask, j0, P, vec_ok));
    before = (int)wave_sum((float)before);               // counts < 2^24: exact in fp32
    if (lane == 0) wsum[wv] = before;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; carry = t; }
    __syncthreads();
    // ---- this segment: 16 cells per thread
    const int i0 = seg0 + tid * SCAN_CPT;
    const unsigned int bits = i0 < P ? kept16(roi_mask, pad_mask, i0, P, vec_ok) : 0u;
    const int cnt = __popc(bits);
    int sc = cnt;                                        // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc, o, 64); if (lane >= o) sc += t; }
    __syncthreads();                                     // wsum is reused
    if (lane == 63) wsum[wv] = sc;
    __syncthreads();
    int off = carry + sc - cnt;
    for (int k = 0; k < wv; ++k) off += wsum[k];
#pragma unroll
    for (int c = 0; c < SCAN_CPT / 4; ++c) {
        const int j0 = i0 + 4 * c;
        int v4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool f = (bits >> (4 * c + k)) & 1u;
            v4[k] = f ? off : -1;
            if (f) { s2pos[off] = j0 + k; ++off; }
        }
        if (vec_ok && j0 + 3 < P) *reinterpret_cast<int4*>(pos2s + j0) = make_int4(v4[0], v4[1], v4[2], v4[3]);
        else for (int k = 0; k < 4; ++k) if (j0 + k < P) pos2s[j0 + k] = v4[k];
    }
    if (blk == nblk - 1 && tid == 1023) *S_out = off;          // the last thread of the last segment ends at the total
}

__global__ __launch_bounds__(1024) void csr_scan_positions_kernel(const unsigned char* __restrict__ roi_mask, const unsigned char* __restrict__ pad_mask,
                                                                  int* __restrict__ pos2s, int* __restrict__ s2pos, int* __restrict__ S_out, int P) {
    csr_scan_positions_block(blockIdx.x, gridDim.x, roi_mask, pad_mask, pos2s, s2pos, S_out, P);
}

// per query: OR the rects of (self + matched RoIs) into an LDS bitmask over P cells, drop cells that are not
// in the key list (padding), write the bitmask and the row count.
// (a batch of samples: the keys of a query lie in its own sample's Pg = V*h*w cells, so the bitmask only covers the 32-cell words
//  [w0, w0 + nwords) that overlap them, w0 = sample * Pg / 32)
__global__ __launch_bounds__(256) void csr_count_kernel(const int* __restrict__ rect, const int* __restrict__ match, const int* __restrict__ pos2s,
                                                        unsigned int* __restrict__ bits, int* __restrict__ row_count, int h, int w, int V, int topk, int nwords) {
    extern __shared__ unsigned int sb[];
    __shared__ int cnt;
    __shared__ int rfirst[256], rnw[256];         // per listed rect: its first cell, cells per row
    __shared__ int rstart[257];                   // items before rect j of the chunk (an item = one cell of one rect)
    __shared__ int wtot[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w0 = (int)(((long long)(rect[r * 5] / V) * V * h * w) >> 5);
    for (int i = tid; i < nwords; i += 256) sb[i] = 0u;
    if (tid == 0) cnt = 0;
    const int nm = 1 + V * topk;
    // The T path lists 1 + 12 * 20 candidates per query, nearly all of them -1: a loop over them is 241 dependent match -> rect -> key-list
    // round trips (67 us per launch).  Here the candidates are fetched 256 at a time, one per thread, their sizes scanned, and the cells of
    // all listed rects are walked as one flat item list with the key-list reads of a pass in flight together.
    for (int base = 0; base < nm; base += 256) {
        int n = 0, first = 0, nw = 1;
        const int j = base + tid;
        if (j < nm) {
            const int m = (j == 0) ? r : match[(long long)r * V * topk + (j - 1)];
            if (m >= 0) {
                const int v = rect[m * 5], y0 = rect[m * 5 + 1], y1 = rect[m * 5 + 2], x0 = rect[m * 5 + 3], x1 = rect[m * 5 + 4];
                if (y1 >= y0 && x1 >= x0) { nw = x1 - x0 + 1; n = (y1 - y0 + 1) * nw; first = (v * h + y0) * w + x0; }
            }
        }
        int s_ = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s_, o, 64); if (lane >= o) s_ += t; }
        __syncthreads();                             // (the previous chunk's readers of rstart / rfirst are done; sb is zeroed)
        if (lane == 63) wtot[wv] = s_;
        __syncthreads();
        int off = 0;
        for (int q = 0; q < wv; ++q) off += wtot[q];
        rfirst[tid] = first; rnw[tid] = nw;
        rstart[tid + 1] = off + s_;
        if (tid == 0) rstart[0] = 0;
        __syncthreads();
        const int total = rstart[256];
        for (int i0 = 0; i0 < total; i0 += 256 * 4) {
            int pos[4];
            bool on[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int it = i0 + 256 * k + tid;
                pos[k] = 0; on[k] = false;
                if (it < total) {
                    int lo = 0, hi = 255;              // the rect of item `it`: last j with rstart[j] <= it
#pragma unroll
                    for (int stp = 0; stp < 8; ++stp) { const int mid = (lo + hi + 1) >> 1; if (rstart[mid] <= it) lo = mid; else hi = mid - 1; }
                    const int i = it - rstart[lo], nwl = rnw[lo];
                    pos[k] = rfirst[lo] + (i / nwl) * w + i % nwl;
                    on[k] = pos2s[pos[k]] >= 0;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (on[k]) atomicOr(&sb[(pos[k] >> 5) - w0], 1u << (pos[k] & 31));
        }
    }
    __syncthreads();
    int c = 0;
    for (int i = tid; i < nwords; i += 256) { const unsigned int x = sb[i]; bits[(long long)r * nwords + i] = x; c += __popc(x); }
    c = (int)wave_sum((float)c);   // counts < 2^24: exact in fp32
    if ((tid & 63) == 0) atomicAdd(&cnt, c);
    __syncthreads();
    if (tid == 0) row_count[r] = cnt;
}

// per query: row_ptr[r] = sum(row_count[0..r)), then expand the bitmask into ascending key indices.
__global__ __launch_bounds__(256) void csr_fill_kernel(const unsigned int* __restrict__ bits, const int* __restrict__ row_count, const int* __restrict__ pos2s,
                                                       const int* __restrict__ rect, int* __restrict__ row_ptr, int* __restrict__ col_idx,
                                                       int* __restrict__ nnz_out, int R, int nwords, int cells_per_sample, int V, int col_cap) {
    __shared__ int sbase;
    __shared__ int wsum[4];
    __shared__ int carry;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int part = 0;
    for (int i = tid; i < r; i += 256) part += row_count[i];
    part = (int)wave_sum((float)part);
    if (tid == 0) { sbase = 0; carry = 0; }
    __syncthreads();
    if (lane == 0) atomicAdd(&sbase, part);
    __syncthreads();
    const int base = sbase;
    if (tid == 0) {
        row_ptr[r] = base;
        if (r == R - 1) { row_ptr[R] = base + row_count[r]; nnz_out[0] = base + row_count[r]; }
        if (base + row_count[r] > col_cap) nnz_out[1] = 1;          // overflow flag: col_idx capacity exceeded
    }
    const int wbase = (int)(((long long)(rect[r * 5] / V) * cells_per_sample) >> 5);
    for (int w0 = 0; w0 < nwords; w0 += 256) {
        const int wi = w0 + tid;
        const unsigned int x = wi < nwords ? bits[(long long)r * nwords + wi] : 0u;
        const int c = __popc(x);
        // inclusive wave scan of c
        int s = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        int off = carry + s - c;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        unsigned int y = x;
        while (y) {
            const int bit = __ffs(y) - 1;
            y &= y - 1;
            if (base + off < col_cap) col_idx[base + off] = pos2s[(wbase + wi) * 32 + bit];
            ++off;
        }
        __syncthreads();
        if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// S-path CSR over the RoI-feature memory (R*49 rows): keys of query r = 49 cells of self, then of each kept
// correlated RoI in (view, rank) order (RH/mv2d_s_head.py:184-192 + box_correlation.py:165-193).
// Blocks of CFC_ROWS rows (round 4; round 1-3: ONE block walked all rows, 56 us at 4800 rows): a block first counts the RoIs listed by all
// earlier rows itself (a few KB of match entries, cheaper than a second launch or a look-back chain), scans its own rows, then every wave
// writes whole rows -- lanes = consecutive entries, the RoI ids of the row compacted through 256 B of LDS.
constexpr int CFC_ROWS = 256;
__device__ __forceinline__ void csr_from_corr_block(int blk, const int* __restrict__ match, int* __restrict__ row_ptr, int* __restrict__ col_idx,
                                                    int* __restrict__ nnz_out, int R, int nm /* V * topk */) {
    __shared__ int wsum[16];
    __shared__ int before_s;
    __shared__ int rowoff[CFC_ROWS + 1];
    __shared__ int ids[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = blk * CFC_ROWS;
    // ---- RoIs listed by the rows before this block (own RoI + kept matches)
    int before = 0;
    for (long long e = tid; e < (long long)base * nm; e += 1024) before += match[e] >= 0 ? 1 : 0;
    before = (int)wave_sum((float)before);               // counts < 2^24: exact in fp32
    if (lane == 0) wsum[wv] = before;
    __syncthreads();
    if (tid == 0) { int t = base; for (int k = 0; k < 16; ++k) t += wsum[k]; before_s = t; }
    __syncthreads();
    // ---- this block's rows: one thread per row counts, the first four waves scan
    int n = 0;
    const int r_own = base + tid;
    if (tid < CFC_ROWS && r_own < R) { n = 1; for (int j = 0; j < nm; ++j) n += match[(long long)r_own * nm + j] >= 0 ? 1 : 0; }
    int sc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc, o, 64); if (lane >= o) sc += t; }
    __syncthreads();                                     // wsum is reused
    if (lane == 63) wsum[wv] = sc;
    __syncthreads();
    if (tid < CFC_ROWS) {
        int off = before_s + sc - n;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        rowoff[tid] = off;
        if (r_own < R) row_ptr[r_own] = off * 49;
        if (r_own == R - 1) { row_ptr[R] = (off + n) * 49; *nnz_out = (off + n) * 49; }
    }
    __syncthreads();
    // ---- a wave per row: compact the row's RoI ids (slot 0 = the row itself), then write 49 consecutive cells per id
    for (int q = wv; q < CFC_ROWS; q += 16) {
        const int r = base + q;
        if (r >= R) break;
        // list positions p = 0 (the row itself), 1 .. nm (its match entries), 64 at a time (one pass when nm < 64: the shipped 'topk_matched' lists;
        // 'all_matched' lists up to views x 128 RoIs, round 6)
        int* out = col_idx + (long long)rowoff[q] * 49;
        for (int p0 = 0; p0 <= nm; p0 += 64) {
            const int p = p0 + lane;
            int id = -1;
            if (p == 0) id = r;
            else if (p <= nm) id = match[(long long)r * nm + p - 1];
            const unsigned long long bal = __ballot(id >= 0);
            if (id >= 0) ids[wv][__popcll(bal & ((1ull << lane) - 1ull))] = id;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int cnt = __popcll(bal) * 49;
            for (int e = lane; e < cnt; e += 64) { const int k = e / 49; out[e] = ids[wv][k] * 49 + (e - k * 49); }
            out += cnt;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ __launch_bounds__(1024) void csr_from_corr_kernel(const int* __restrict__ match, int* __restrict__ row_ptr, int* __restrict__ col_idx,
                                                             int* __restrict__ nnz_out, int R, int nm) {
    csr_from_corr_block(blockIdx.x, match, row_ptr, col_idx, nnz_out, R, nm);
}

// S path, launch order of the attention blocks (xattn_tile_kernel's `order`): the queries of a sample ranked by the SMALLEST RoI they list (own
// RoI or a matched one) -- RoIs of different views that are matched with each other then run side by side on one XCD and share its L2
// (round 4: the overlapping-rig workload re-fetched 1.86 x its distinct key rows from HBM in the natural order).  Same ranking as
// query_order_kernel (xattn_order.hip), with the key taken straight from the correlation lists: no CSR needed, so it rides in the same
// launch.  Groups: sample g = rows [grp_start[g], grp_start[g+1]); the bucket-padding rows behind the last sample keep their places.
constexpr int SORD_MAX = 4096, SORD_CHUNK = 128;
__device__ __forceinline__ void s_order_block(int g, int chunk, const int* __restrict__ match, const int* __restrict__ grp_start, int n_grp, int R, int nm,
                                              int* __restrict__ perm, int* __restrict__ flags) {
    __shared__ int key[SORD_MAX];
    const int tid = threadIdx.x;
    const int lo = g < n_grp ? grp_start[g] : grp_start[n_grp];
    const int hi = g < n_grp ? grp_start[g + 1] : R;
    const int n = hi - lo;
    if (n > SORD_MAX || g >= n_grp) {                          // too many queries for the LDS ranking / padding rows: natural order
        if (chunk == 0 && tid == 0 && flags && g < n_grp) flags[0] = 1;      // (like query_order_kernel: a sample whose L2-sharing order was dropped says so)
        if (chunk == 0) for (int i = tid; i < n; i += 1024) perm[lo + i] = lo + i;
        return;
    }
    if (chunk * SORD_CHUNK >= n) return;
    // Two orders (round 5).  Matched RoIs are common (overlapping rig: 4 RoIs per query): by the smallest listed RoI, so that the queries that read
    // the same RoIs run side by side and share an L2 (any length-aware reordering, even inside windows of 64 places, costs more than it balances:
    // 173 -> 206 us per layer).  Matched RoIs are rare (ring rig: 7 % of the queries list a second RoI, nothing to share): rows that list MORE RoIs
    // first, then the smallest RoI -- the one-launch cross attention gives a block 6-8 consecutive queries, one per wave, and waits for the longest,
    // so rows of equal length belong into the same blocks (131 -> 124 us per layer; mixed blocks waited 1.75 x in 40 % of the cases).  The choice
    // is made per sample from its own correlation lists (fewer than one matched RoI per two queries -> length first), identically by every block.
    __shared__ int n_matched;
    if (tid == 0) n_matched = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n; i += 1024) {
        int k = lo + i, nc = 1;
        for (int j = 0; j < nm; ++j) { const int m = match[(long long)(lo + i) * nm + j]; if (m >= 0) { k = min(k, m); ++nc; } }
        key[i] = ((63 - min(nc, 63)) << 24) | (k & 0xffffff);
        mine += nc - 1;
    }
    if (mine) atomicAdd(&n_matched, mine);
    __syncthreads();
    const int kmask = (2 * n_matched < n) ? 0x7fffffff : 0x00ffffff;      // length-major, or the smallest RoI alone
    const int i = chunk * SORD_CHUNK + (tid >> 3), sub = tid & 7;
    int rank = 0;
    if (i < n) {
        const int ki = key[i] & kmask;
        for (int j = sub; j < n; j += 8) { const int kj = key[j] & kmask; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    rank += __shfl_xor(rank, 4);
    if (i < n && sub == 0) perm[lo + rank] = lo + i;
}

// S path: the scan of the RoI-tap positions, the CSR over the correlated RoIs and the launch order of the attention blocks are independent:
// one launch, block roles by index
__global__ __launch_bounds__(1024) void scan_and_csr_kernel(const unsigned char* __restrict__ roi_mask, const unsigned char* __restrict__ pad_mask,
                                                            int* __restrict__ pos2s, int* __restrict__ s2pos, int* __restrict__ S_out, int P, int nscan,
                                                            const int* __restrict__ match, int* __restrict__ row_ptr, int* __restrict__ col_idx,
                                                            int* __restrict__ nnz_out, int R, int nm, int ncsr, const int* __restrict__ grp_start, int n_grp,
                                                            int ord_chunks, int* __restrict__ perm, int* __restrict__ order_flags) {
    const int b = blockIdx.x;
    if (b < nscan) csr_scan_positions_block(b, nscan, roi_mask, pad_mask, pos2s, s2pos, S_out, P);
    else if (b < nscan + ncsr) csr_from_corr_block(b - nscan, match, row_ptr, col_idx, nnz_out, R, nm);
    else s_order_block((b - nscan - ncsr) / ord_chunks, (b - nscan - ncsr) % ord_chunks, match, grp_start, n_grp, R, nm, perm, order_flags);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Frustum rows of the index-exact route's PE block, fp32 [S, 3 D], and nothing else (round 5; pe_inputs_kernel<true> also writes the key16
// rows, a second fp32 logarithm and the feature rows nobody reads on that route, and spends ~200 fp64 operations per value on two IEEE
// divisions and the library log: 156 us per 141 k positions, VALU-bound).  Same arithmetic in fp64 -- MU/pe.py:96-131: point of depth bin d
// through img2lidar, normalisation to the position range, inverse_sigmoid (clamp(0,1), both arguments clamped at 1e-5), then .float() --
// regrouped so that it costs ~40 operations per value:
//   * the point is linear in the depth, M (cw dm, ch dm, d, 1) = dm (M0 cw + M1 ch) + d M2 + M3 (the per-position part is wave-uniform);
//   * the normalisation multiplies by 1 / range;
//   * log(x1 / x2) = log x1 - log x2 with a table-driven logarithm: x = 2^e m, c = the centre of m's 1/128 bin, r = m / c - 1 (|r| <= 2^-8),
//     log m = log c + r - r^2/2 + ... - r^6/6 (truncation 2e-18); the table holds 1 / c and log c in fp64 (filled on the host at first use).
// Every regrouping moves the fp64 value by a few 1e-16, i.e. the fp32 result differs from pe_inputs_kernel<true>'s in about one element
// per 1e8, by one ulp (tests/test_gpu_kernels.py::test_pe_frustum_rows_fast_equals_reference_order).
// ------------------------------------------------------------------------------------------------------------------------------
struct LogTab { double t[256]; };         // [2 i] = 1 / c_i, [2 i + 1] = log c_i, c_i = 1 + (i + 0.5) / 128; a kernel ARGUMENT (2 KB): no per-device symbol, no copy inside a graph capture

__device__ __forceinline__ double log_diff_tab(double x1, double x2, const double* __restrict__ tab) {
    const long long b1 = __double_as_longlong(x1), b2 = __double_as_longlong(x2);
    const int e1 = (int)((b1 >> 52) & 0x7ff), e2 = (int)((b2 >> 52) & 0x7ff);
    const int i1 = (int)((b1 >> 45) & 127), i2 = (int)((b2 >> 45) & 127);
    const double m1 = __longlong_as_double((b1 & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    const double m2 = __longlong_as_double((b2 & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    const double r1 = fma(m1, tab[2 * i1], -1.0), r2 = fma(m2, tab[2 * i2], -1.0);
    auto l1p = [](double r) {
        double p = fma(r, -1.0 / 6.0, 1.0 / 5.0);
        p = fma(r, p, -0.25);
        p = fma(r, p, 1.0 / 3.0);
        p = fma(r, p, -0.5);
        p = fma(r, p, 1.0);
        return r * p;
    };
    return fma((double)(e1 - e2), 0.6931471805599453094, (tab[2 * i1 + 1] - tab[2 * i2 + 1]) + (l1p(r1) - l1p(r2)));
}

__global__ __launch_bounds__(256) void pe_frustum_f32_kernel(const int* __restrict__ s2pos, const int* __restrict__ S_dev, const double* __restrict__ img2lidar,
                                                             const double* __restrict__ coords_w, const double* __restrict__ coords_h,
                                                             const double* __restrict__ coords_d, float* __restrict__ out, int h, int w, int D,
                                                             double pr0, double pr1, double pr2, double ipd0, double ipd1, double ipd2, LogTab lt) {
    __shared__ double tab[256];
    tab[threadIdx.x] = lt.t[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = *S_dev;
    for (int s = blockIdx.x * 4 + wave; s < S; s += gridDim.x * 4) {
        const int pos = __builtin_amdgcn_readfirstlane(s2pos[s]);
        const int v = pos / (h * w), rem = pos - v * h * w, y = rem / w, x = rem - y * w;
        const double* M = img2lidar + v * 16;
        const double cw = coords_w[x], chh = coords_h[y];
        const double u[3] = {fma(M[0], cw, M[1] * chh), fma(M[4], cw, M[5] * chh), fma(M[8], cw, M[9] * chh)};
        const double m2[3] = {M[2], M[6], M[10]}, m3[3] = {M[3] - pr0, M[7] - pr1, M[11] - pr2};
        const double ipd[3] = {ipd0, ipd1, ipd2};
        for (int dk = lane; dk < D; dk += 64) {
            const double d = coords_d[dk];
            const double dm = d < 1e-3 ? 1e-3 : d;
            float o[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double n = fma(u[i], dm, fma(m2[i], d, m3[i])) * ipd[i];
                n = n < 0.0 ? 0.0 : (n > 1.0 ? 1.0 : n);
                const double x1 = n < 1e-5 ? 1e-5 : n;
                const double x2 = (1.0 - n) < 1e-5 ? 1e-5 : (1.0 - n);
                o[i] = (float)log_diff_tab(x1, x2, tab);
            }
            float* dst = out + (long long)s * (3 * D) + dk * 3;
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a2 inputs: for every key position s of the compacted list build the three PE-MLP input rows and gather
// the feature row (MU/pe.py:84-135 frustum coords in fp64, MU/positional_encoding.py:78-95 sine features).
// ------------------------------------------------------------------------------------------------
// One wave per key position (4 per block): lane l moves channels 4l..4l+3 of the feature row with 16-byte accesses, computes depth
// bin l of the frustum row (3 coordinates) and 6 sine channels; the two key16 (fp16) input rows (384 B, 768 B) are assembled in LDS and
// written with 16-byte stores.  (Round 1: one block per position with 2- and 4-byte accesses ran at 2 TB/s of its 130 MB.)
template <bool EXACT>      // EXACT: also the unrounded fp32 rows of the engine's index-exact validation mode (fp64 log, library sin / cos)
__global__ __launch_bounds__(256) void pe_inputs_kernel(const int* __restrict__ s2pos, const int* __restrict__ S_dev, const float* __restrict__ featcl,
                                                        const double* __restrict__ img2lidar, const double* __restrict__ coords_w, const double* __restrict__ coords_h,
                                                        const double* __restrict__ coords_d, const float* __restrict__ embeds, const float* __restrict__ dim_t,
                                                        unsigned short* __restrict__ A_frustum, unsigned short* __restrict__ A_sine, unsigned short* __restrict__ Xf_k16,
                                                        float* __restrict__ Xf_f32, float* __restrict__ A_frustum_f32, float* __restrict__ A_sine_f32,
                                                        int h, int w, int P, int D, double pr0, double pr1, double pr2,
                                                        double pd0, double pd1, double pd2) {
    __shared__ __attribute__((aligned(16))) unsigned short rowbuf[4][3 * 256 + 384];      // frustum row (<= 768 values) | sine row (384)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave;
    if (s >= *S_dev) return;                                  // (whole waves leave; no block barrier below)
    // (the position is the same for all lanes of the wave: made uniform explicitly, so that the per-view matrix and the column / row
    // coordinates are scalar loads instead of 14 broadcast vector loads per lane)
    const int pos = __builtin_amdgcn_readfirstlane(s2pos[s]);
    const int v = pos / (h * w), rem = pos - v * h * w, y = rem / w, x = rem - y * w;
    unsigned short* fr_row = rowbuf[wave];
    unsigned short* si_row = rowbuf[wave] + 3 * 256;
    // feature row gather (fp32 kept for the K = feat + pe sum, key16 for the SE gate and the value rows)
    {
        const float4 f = *reinterpret_cast<const float4*>(featcl + (long long)pos * C + 4 * lane);
        if (Xf_f32) *reinterpret_cast<float4*>(Xf_f32 + (long long)s * C + 4 * lane) = f;
        *reinterpret_cast<uint2*>(Xf_k16 + (long long)s * C + 4 * lane) = make_uint2(pack_k16x2(f.x, f.y), pack_k16x2(f.z, f.w));
    }
    if constexpr (EXACT) {
        for (int dk = lane; dk < D; dk += 64) {
            const double d = coords_d[dk];
            const double dm = d < 1e-3 ? 1e-3 : d;
            const double p[4] = {coords_w[x] * dm, coords_h[y] * dm, d, 1.0};
            const double* M = img2lidar + v * 16;
            const double pr[3] = {pr0, pr1, pr2}, pd[3] = {pd0, pd1, pd2};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = acc + M[i * 4 + k] * p[k];
                double n = (acc - pr[i]) / pd[i];
                n = n < 0.0 ? 0.0 : (n > 1.0 ? 1.0 : n);          // inverse_sigmoid: clamp(0,1)
                const double x1 = n < 1e-5 ? 1e-5 : n;
                const double x2 = (1.0 - n) < 1e-5 ? 1e-5 : (1.0 - n);
                fr_row[dk * 3 + i] = f32_to_k16(logf((float)x1 / (float)x2));
                // index-exact route: the unrounded fp32 row, every step in fp64 and in the reference's operation order (MU/pe.py:119-130)
                A_frustum_f32[(long long)s * (3 * D) + dk * 3 + i] = (float)log(x1 / x2);
            }
        }
    } else {
        // default route (round 4): the point of depth bin d is linear in d -- M (cw dm, ch dm, d, 1) = dm (M0 cw + M1 ch) + d M2 + M3 -- so the
        // per-position part is hoisted (wave-uniform) and a coordinate costs two fp64 FMAs; the normalisation multiplies by 1 / range.  (Before:
        // 4 fp64 products + sums and an fp64 DIVISION per coordinate, 192 per position; 96.8 -> 88.6 us per 140 k positions.)  The fp64 result
        // moves by ~1e-16 relative; it is rounded to fp32 for the quotient / logarithm and to key16 right after.
        const double* M = img2lidar + v * 16;
        const double cw = coords_w[x], chh = coords_h[y];
        const double u[3] = {fma(M[0], cw, M[1] * chh), fma(M[4], cw, M[5] * chh), fma(M[8], cw, M[9] * chh)};
        const double m2[3] = {M[2], M[6], M[10]}, m3[3] = {M[3] - pr0, M[7] - pr1, M[11] - pr2};
        const double ipd[3] = {1.0 / pd0, 1.0 / pd1, 1.0 / pd2};
        for (int dk = lane; dk < D; dk += 64) {
            const double d = coords_d[dk];
            const double dm = d < 1e-3 ? 1e-3 : d;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double n = fma(u[i], dm, fma(m2[i], d, m3[i])) * ipd[i];
                n = n < 0.0 ? 0.0 : (n > 1.0 ? 1.0 : n);          // inverse_sigmoid: clamp(0,1)
                const double x1 = n < 1e-5 ? 1e-5 : n;
                const double x2 = (1.0 - n) < 1e-5 ? 1e-5 : (1.0 - n);
                // quotient and logarithm in fp32: 1e-7 absolute against a value that is rounded to key16 (fp16) right here
                fr_row[dk * 3 + i] = f32_to_k16(logf((float)x1 / (float)x2));
            }
        }
    }
    // sine features, channel order (n | y | x).  NOT interleaved: the reference stacks sin/cos on dim=4 of a
    // 5-D tensor (MU/positional_encoding.py:86-94), so within an axis channels 0..63 = sin(e / dim_t[2j]) and
    // channels 64..127 = cos(e / dim_t[2j+1]).
    if (!A_sine) {                                            // the sine branch comes from the engine's folded table: only the frustum row is needed
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int nf0 = 3 * D / 8;
        for (int q = lane; q < nf0; q += 64)
            *reinterpret_cast<uint4*>(A_frustum + (long long)s * (3 * D) + 8 * q) = *reinterpret_cast<const uint4*>(fr_row + 8 * q);
        return;
    }
    const float en = embeds[pos], ey = embeds[P + pos], ex = embeds[2 * P + pos];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int ch = lane + 64 * k;
        const int axis = ch >> 7, i = ch & 127;
        const float e = axis == 0 ? en : (axis == 1 ? ey : ex);
        // arguments lie in [0, 2 pi]: the hardware sin / cos (v_sin_f32 on x / 2 pi, ~1e-6 absolute) is as good as the library call
        // for a value that is rounded to key16 (fp16) right here
        const float a = e / dim_t[i < 64 ? 2 * i : 2 * (i - 64) + 1];
        si_row[ch] = f32_to_k16(i < 64 ? __sinf(a) : __cosf(a));
        if (EXACT) A_sine_f32[(long long)s * 384 + ch] = i < 64 ? sinf(a) : cosf(a);
    }
    __builtin_amdgcn_wave_barrier();                          // the rows are read back by the same wave only
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): LDS writes landed
    const int nf = 3 * D / 8;                                 // 16-byte chunks of the frustum row (D % 8 == 0)
    for (int q = lane; q < nf; q += 64)
        *reinterpret_cast<uint4*>(A_frustum + (long long)s * (3 * D) + 8 * q) = *reinterpret_cast<const uint4*>(fr_row + 8 * q);
    if (lane < 48) *reinterpret_cast<uint4*>(A_sine + (long long)s * 384 + 8 * lane) = *reinterpret_cast<const uint4*>(si_row + 8 * lane);
}

// ------------------------------------------------------------------------------------------------
// a21: NMS-free decode (CB/coders/nms_free_coder.py:49-102, CB/util.py:60-87,
//      cross_attention_head.py:372): single block, bitonic sort of (logit, index) in LDS.
//      Order: logit descending, lower flat index first among equals.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void decode_topk_kernel(const float* __restrict__ cls, const float* __restrict__ reg, int R, int ncls, int max_num,
                                                           int npow2, float r0, float r1, float r2, float r3, float r4, float r5,
                                                           float* __restrict__ boxes, float* __restrict__ scores, long long* __restrict__ labels,
                                                           long long* __restrict__ bbox_index, int* __restrict__ count_out,
                                                           long long* __restrict__ topk_index_dbg, const int* __restrict__ grp_start,
                                                           float* __restrict__ payload) {
    // keys: 64-bit (monotone logit bits << 32 | ~index) -> all distinct, "larger" = higher logit, then LOWER index.
    // 1) 8 rounds of 8-bit radix select find the K-th largest key; 2) the K survivors are ranked by counting
    //    (K^2 compares spread over 1024 threads) — no full sort of the R*ncls candidates.
    extern __shared__ unsigned long long keys[];               // [n] all keys, then [1024] survivors
    __shared__ int hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ int want_s, nsel;
    __shared__ int kept_off[1025];
    // one block per sample of the batch: rows [grp_start[g], grp_start[g+1]), outputs [g][max_num] (bbox_index relative to the sample)
    if (grp_start) {
        const int g = blockIdx.x, gs = grp_start[g];
        R = grp_start[g + 1] - gs;
        cls += (long long)gs * ncls; reg += (long long)gs * 10;
        boxes += (long long)g * max_num * 9; scores += (long long)g * max_num; labels += (long long)g * max_num;
        bbox_index += (long long)g * max_num; count_out += g;
        if (topk_index_dbg) topk_index_dbg += (long long)g * max_num;
        if (payload) payload += (long long)g * ((long long)max_num * 11 + 1);
    }
    const int tid = threadIdx.x, n = R * ncls;
    const int K = min(max_num, n);
    unsigned long long* sel = keys + npow2;
    for (int i = tid; i < n; i += 1024) {
        unsigned int u = __float_as_uint(cls[i]);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);           // monotone float -> uint
        keys[i] = ((unsigned long long)u << 32) | (unsigned int)(0xffffffffu - (unsigned int)i);
    }
    if (tid == 0) { prefix_s = 0ull; want_s = K; nsel = 0; }
    __syncthreads();
    // radix select over the 32 logit bits (4 rounds); the bin holding the K-th element is found by one wave with a
    // suffix scan over 256 bins (4 bins per lane)
    for (int shift = 56; shift >= 32; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        const unsigned long long mask_hi = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long k = keys[i];
            if ((k & mask_hi) == prefix) atomicAdd(&hist[(int)((k >> shift) & 0xffull)], 1);
        }
        __syncthreads();
        if (tid < 64) {
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const int s4 = h0 + h1 + h2 + h3;
            int suf = s4;                                   // inclusive suffix sum over lanes tid..63
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(suf, o, 64); if (tid + o < 64) suf += t; }
            const int want = want_s;
            const unsigned long long bal = __ballot(suf >= want);
            const int hit = 63 - __clzll(bal);              // highest lane whose suffix still covers `want`
            if (tid == hit) {
                int w = want - (suf - s4);                  // still wanted inside this lane's 4 bins
                int b = 4 * tid + 3;
                const int hh[4] = {h0, h1, h2, h3};
                for (int e = 3; e > 0; --e) { if (hh[e] >= w) break; w -= hh[e]; --b; }
                prefix_s = prefix | ((unsigned long long)b << shift);
                want_s = w;
            }
        }
        __syncthreads();
    }
    const unsigned long long kth_hi = prefix_s;                   // upper 32 bits of the K-th largest key
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long k = keys[i];
        if ((k & 0xffffffff00000000ull) >= kth_hi) { const int slot = atomicAdd(&nsel, 1); if (slot < 1024) sel[slot] = k; }
    }
    __syncthreads();
    // rank by counting among the survivors (K plus, rarely, a few equal-logit candidates; ties resolved by the low
    // 32 bits = lower index first); ranks >= K are dropped
    const int ns = min(nsel, 1024);
    unsigned long long mine = 0ull;
    int rank = 1 << 30;
    if (tid < ns) {
        mine = sel[tid];
        rank = 0;
        for (int j = 0; j < ns; ++j) rank += sel[j] > mine ? 1 : 0;
    }
    __syncthreads();
    if (rank < K) keys[rank] = mine;                              // keys[0..K) now sorted descending
    __syncthreads();
    // ---- gather + denormalise + centre-range filter, kept entries keep their rank order
    int keep = 0;
    float bx[9]; float sc = 0.f; int idx = 0;
    if (tid < K) {
        idx = (int)(0xffffffffu - (unsigned int)(keys[tid] & 0xffffffffull));
        if (topk_index_dbg) topk_index_dbg[tid] = idx;
        const int q = idx / ncls;
        const float* bp = reg + (long long)q * 10;
        sc = 1.0f / (1.0f + expf(-cls[idx]));
        bx[0] = bp[0]; bx[1] = bp[1]; bx[2] = bp[4];
        bx[3] = expf(bp[2]); bx[4] = expf(bp[3]); bx[5] = expf(bp[5]);
        bx[6] = atan2f(bp[6], bp[7]); bx[7] = bp[8]; bx[8] = bp[9];
        keep = (bx[0] >= r0 && bx[1] >= r1 && bx[2] >= r2 && bx[0] <= r3 && bx[1] <= r4 && bx[2] <= r5) ? 1 : 0;
    }
    // exclusive scan of the keep flags over 1024 threads (wave ballots)
    {
        const int lane = tid & 63, wv = tid >> 6;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) hist[wv] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int k = 0; k < wv; ++k) off += hist[k];
        kept_off[tid] = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (tid == 1023) { *count_out = kept_off[tid] + keep; }
    }
    if (keep) {
        const int o = kept_off[tid];
        bx[2] = bx[2] - bx[5] * 0.5f;                       // gravity centre -> bottom centre (:372)
        for (int i = 0; i < 9; ++i) boxes[o * 9 + i] = bx[i];
        scores[o] = sc;
        labels[o] = idx % ncls;
        bbox_index[o] = idx / ncls;
    }
    // optional: the sample's row of the wire format of the per-step all-gather (pack_detections_kernel's output) in the same launch:
    // max_num rows of (box[9], score, label) as fp32, rows >= count zeroed, then the count
    if (payload) {
        __syncthreads();                                     // the block's own global writes above are visible to all of its threads
        const int cnt = min(*count_out, max_num);
        for (int i = tid; i < max_num * 11; i += 1024) {
            const int r = i / 11, c = i - r * 11;
            float v = 0.f;
            if (r < cnt) v = c < 9 ? boxes[r * 9 + c] : (c == 9 ? scores[r] : (float)labels[r]);
            payload[i] = v;
        }
        if (tid == 0) payload[(long long)max_num * 11] = (float)*count_out;
    }
}

// ------------------------------------------------------------------------------------------------
// f1 ("next" row): the caller's post-decoder step (mmdet3d_plugin/models/detectors/mv2d.py:265-287): scores scattered to
// [K, num_classes+1], mmdet3d box3d_multiclass_nms with nms_thr = 1.0 (rotated IoU can never exceed 1 -> nothing is suppressed):
// per class, boxes with score > score_thr in descending score order, classes concatenated in ascending order; only if more than
// max_num survive, the max_num best by score (then globally score-descending).  Single block, rank by counting (n <= 1024).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void result_pack_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const long long* __restrict__ labels,
                                                           const int* __restrict__ count, float score_thr, int max_num, float* __restrict__ out_boxes,
                                                           float* __restrict__ out_scores, long long* __restrict__ out_labels, int* __restrict__ out_count,
                                                           int in_stride) {
    __shared__ float ss[1024];
    __shared__ int sl[1024];
    __shared__ int nkeep;
    {   // one block per sample of the batch
        const long long g = blockIdx.x;
        boxes += g * in_stride * 9; scores += g * in_stride; labels += g * in_stride; count += g;
        out_boxes += g * max_num * 9; out_scores += g * max_num; out_labels += g * max_num; out_count += g;
    }
    const int tid = threadIdx.x, n = min(*count, 1024);
    const bool have = tid < n;
    const float sc = have ? scores[tid] : 0.f;
    const int lb = have ? (int)labels[tid] : 0;
    const bool keep = have && sc > score_thr;
    ss[tid] = keep ? sc : -1.f;
    sl[tid] = keep ? lb : 1 << 20;
    if (tid == 0) nkeep = 0;
    __syncthreads();
    if (keep) atomicAdd(&nkeep, 1);
    __syncthreads();
    const int total = nkeep;
    if (!keep) { if (tid == 0) *out_count = min(total, max_num); return; }
    int rank = 0;
    if (total <= max_num) {
        for (int j = 0; j < n; ++j) {          // (label asc, score desc, index asc)
            const int lj = sl[j]; const float sj = ss[j];
            rank += (lj < lb || (lj == lb && (sj > sc || (sj == sc && j < tid)))) ? 1 : 0;
        }
    } else {
        for (int j = 0; j < n; ++j) {          // global score desc, index asc
            const float sj = ss[j];
            rank += (sl[j] < (1 << 20) && (sj > sc || (sj == sc && j < tid))) ? 1 : 0;
        }
    }
    if (rank < max_num) {
        for (int i = 0; i < 9; ++i) out_boxes[rank * 9 + i] = boxes[tid * 9 + i];
        out_scores[rank] = sc;
        out_labels[rank] = lb;
    }
    if (tid == 0) *out_count = min(total, max_num);
}

// wire format of the per-step all-gather of decoded boxes (mv2d_amd/dist.py): per sample max_num rows of (box[9], score, label) as
// fp32, rows >= count zeroed, then the count
__global__ __launch_bounds__(256) void pack_detections_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const long long* __restrict__ labels,
                                                              const int* __restrict__ count, float* __restrict__ out, int max_num, int in_stride) {
    const long long g = blockIdx.x;
    const int n = min(count[g], max_num);
    float* o = out + g * ((long long)max_num * 11 + 1);
    for (int i = threadIdx.x; i < max_num * 11; i += 256) {
        const int r = i / 11, c = i - r * 11;
        float v = 0.f;
        if (r < n) v = c < 9 ? boxes[(g * in_stride + r) * 9 + c] : (c == 9 ? scores[g * in_stride + r] : (float)labels[g * in_stride + r]);
        o[i] = v;
    }
    if (threadIdx.x == 0) o[(long long)max_num * 11] = (float)count[g];
}

}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" int mv2d_box_params(const float* rois, const double* viewK, const double* viewE, double* K_roi, float* intr,
                               int ld_intr, float* minv, int R, float roi_size, float intr_scale, float min_size, void* stream) {
    MV2D_CHECK_ARG(rois && viewK && viewE && intr && minv && ld_intr >= 16, "mv2d_box_params: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(box_params_kernel, dim3(cdiv(R, 64)), dim3(64), 0, (hipStream_t)stream, rois, viewK, viewE, K_roi, intr,
                       ld_intr, minv, R, roi_size, intr_scale, min_size);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_refpoint_posemb(const float* center_pred, int ld_cp, const float* minv, const float* dim_t, float* xyz, float* ref,
                                    float* posemb, int R, const float* pc_range, void* stream) {
    MV2D_CHECK_ARG(center_pred && minv && dim_t && xyz && ref && posemb && pc_range, "mv2d_refpoint_posemb: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(refpoint_posemb_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, center_pred, ld_cp, minv, dim_t,
                       xyz, ref, posemb, R, pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0],
                       pc_range[4] - pc_range[1], pc_range[5] - pc_range[2]);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_lidar2img_inverse(const double* K_roi, const double* E, float* minv, int R, void* stream) {
    MV2D_CHECK_ARG(K_roi && E && minv, "mv2d_lidar2img_inverse: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(lidar2img_inverse_kernel, dim3(cdiv(R, 64)), dim3(64), 0, (hipStream_t)stream, K_roi, E, minv, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_posemb3d(const float* ref, const float* dim_t, float* posemb, int R, void* stream) {
    MV2D_CHECK_ARG(ref && dim_t && posemb, "mv2d_posemb3d: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(posemb3d_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, ref, dim_t, posemb, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_roi_align_ex(const float* map0, const float* map1, const float* rois, void* out0, void* out1, float* out0_f32,
                                 float* out1_f32, int R, int H, int W, int channels, float spatial_scale, int sampling_ratio,
                                 const int* map1_index, int out1_is_sum, void* out0_lo, void* out1_lo, void* out0_lo8, void* out1_lo8, int* lo8_flag, void* stream) {
    MV2D_CHECK_ARG(map0 && rois && channels == C, "mv2d_roi_align: needs 256-channel position-major maps");
    MV2D_CHECK_ARG(out0 || out0_f32, "mv2d_roi_align: no output");
    MV2D_CHECK_ARG((!(out0_lo || out0_lo8) || out0) && (!(out1_lo || out1_lo8) || out1), "mv2d_roi_align_ex: a lo output needs its key16 (hi) output");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(roi_align_kernel, dim3(56 * cdiv(R, 8)), dim3(256), 0, (hipStream_t)stream, map0, map1, rois, (unsigned short*)out0,
                       (unsigned short*)out1, out0_f32, out1_f32, H, W, spatial_scale, sampling_ratio, map1_index, out1_is_sum, R,
                       (unsigned short*)out0_lo, (unsigned short*)out1_lo, (unsigned char*)out0_lo8, (unsigned char*)out1_lo8, lo8_flag);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_roi_align(const float* map0, const float* map1, const float* rois, void* out0, void* out1, float* out0_f32,
                              float* out1_f32, int R, int H, int W, int channels, float spatial_scale, int sampling_ratio,
                              const int* map1_index, int out1_is_sum, void* stream) {
    return mv2d_roi_align_ex(map0, map1, rois, out0, out1, out0_f32, out1_f32, R, H, W, channels, spatial_scale, sampling_ratio, map1_index,
                             out1_is_sum, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int mv2d_roi_align_bwd(const float* grad_out, const float* rois, float* grad_map, const int* index, int R, int H, int W,
                                  int channels, float spatial_scale, int sampling_ratio, void* stream) {
    MV2D_CHECK_ARG(grad_out && rois && grad_map && channels == C, "mv2d_roi_align_bwd: needs fp32 [R,49,256] gradients and a 256-channel position-major map");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(R, 7), dim3(256), 0, (hipStream_t)stream, grad_out, rois, grad_map, index, H, W, spatial_scale,
                       sampling_ratio);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_box_correlation(const float* rois, const int* view_start, const double* trans, const float* lin, const float* depths,
                                    int* match, int R, int V, int sample_size, int num_depth, int topk, int pad_h, int pad_w,
                                    float depth_start, float iou_thr, float ratio, int max_per_view, void* stream) {
    MV2D_CHECK_ARG(rois && view_start && trans && lin && depths && match, "mv2d_box_correlation: null pointer");
    MV2D_CHECK_ARG(sample_size * sample_size * num_depth <= 128, "mv2d_box_correlation: sample_size^2*num_depth must be <= 128");
    MV2D_CHECK_ARG(max_per_view <= MAX_PER_VIEW && topk >= 1, "mv2d_box_correlation: too many RoIs in one view (max 1024)");
    if (R == 0) return MV2D_OK;
    const BoxCorrArgs A{rois, view_start, trans, lin, depths, match, V, sample_size, num_depth, topk, (float)(pad_w - 1), (float)(pad_h - 1), depth_start,
                        iou_thr, ratio};
    hipLaunchKernelGGL(box_corr_kernel, dim3(R, V), dim3(128), 0, (hipStream_t)stream, A);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// mv2d_box_params + mv2d_box_correlation + the clearing of `zero_bytes` bytes at zero_ptr (16-byte aligned, a multiple of 16; may be NULL)
// in ONE launch: see frame_geometry_kernel
extern "C" int mv2d_frame_geometry(const float* rois, const double* viewK, const double* viewE, double* K_roi, float* intr, int ld_intr, float* minv,
                                   float roi_size, float intr_scale, float min_size, const int* view_start, const double* trans, const float* lin,
                                   const float* depths, int* match, int R, int V, int sample_size, int num_depth, int topk, int pad_h, int pad_w,
                                   float depth_start, float iou_thr, float ratio, int max_per_view, void* zero_ptr, long long zero_bytes, void* stream) {
    MV2D_CHECK_ARG(rois && viewK && viewE && intr && minv && ld_intr >= 16 && view_start && trans && lin && depths && match, "mv2d_frame_geometry: null pointer");
    MV2D_CHECK_ARG(sample_size * sample_size * num_depth <= 128, "mv2d_frame_geometry: sample_size^2*num_depth must be <= 128");
    MV2D_CHECK_ARG(max_per_view <= MAX_PER_VIEW && topk >= 1, "mv2d_frame_geometry: too many RoIs in one view (max 1024)");
    MV2D_CHECK_ARG(zero_bytes >= 0 && (zero_bytes % 16) == 0 && ((uintptr_t)zero_ptr & 15) == 0 && (zero_ptr || zero_bytes == 0),
                   "mv2d_frame_geometry: the region to clear must be 16-byte aligned and a multiple of 16 bytes");
    if (R == 0) {
        // nothing to launch for, but the frame's mask / flag bytes are still this call's to clear (a direct ABI caller with an empty RoI list
        // would otherwise keep the previous frame's roi_mask and overflow flags)
        if (zero_ptr && zero_bytes > 0 && hipMemsetAsync(zero_ptr, 0, (size_t)zero_bytes, (hipStream_t)stream) != hipSuccess) {
            mv2d_set_error("mv2d_frame_geometry: clearing the zero region failed");
            return MV2D_ERR_LAUNCH;
        }
        return MV2D_OK;
    }
    FrameGeoArgs G{{rois, view_start, trans, lin, depths, match, V, sample_size, num_depth, topk, (float)(pad_w - 1), (float)(pad_h - 1), depth_start,
                    iou_thr, ratio}, viewK, viewE, K_roi, intr, ld_intr, minv, roi_size, intr_scale, min_size, (uint4*)zero_ptr, zero_bytes / 16, R};
    const int nz = (int)((zero_bytes / 16 + GEO_ZERO_PER_BLOCK - 1) / GEO_ZERO_PER_BLOCK);
    hipLaunchKernelGGL(frame_geometry_kernel, dim3(R + cdiv(R, 128) + nz, V), dim3(128), 0, (hipStream_t)stream, G);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// words of the per-query key bitmask: the cells of one sample (V views) plus one word of slack for a window that starts mid-word
static inline int csr_words(int V, int h, int w) { return (V * h * w + 31) / 32 + 1; }

extern "C" long long mv2d_csr_workspace_bytes(int R, int V, int h, int w) {
    return (long long)R * csr_words(V, h, w) * 4;
}

// T-path: roi_mask must be zeroed by the caller (hipMemsetAsync) before this call.
extern "C" int mv2d_mask_compact(const float* rois, const int* match, const unsigned char* pad_mask, unsigned char* roi_mask,
                                 int* rect, int* pos2s, int* s2pos, int* S_out, unsigned int* bits_ws, int* row_count, int* row_ptr,
                                 int* col_idx, int* nnz_out, int col_cap, int R, int V, int h, int w, int topk, float stride,
                                 float expand_stride, int n_samples, void* stream) {
    MV2D_CHECK_ARG(rois && match && pad_mask && roi_mask && rect && pos2s && s2pos && S_out && bits_ws && row_count && row_ptr &&
                       col_idx && nnz_out, "mv2d_mask_compact: null pointer");
    MV2D_CHECK_ARG(R > 0 && n_samples >= 1, "mv2d_mask_compact: R and n_samples must be > 0");
    const int P = n_samples * V * h * w;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(csr_mark_kernel, dim3(R), dim3(64), 0, st, rois, rect, roi_mask, h, w, stride, expand_stride);
    hipLaunchKernelGGL(csr_scan_positions_kernel, dim3(cdiv(P, SCAN_SEG)), dim3(1024), 0, st, roi_mask, pad_mask, pos2s, s2pos, S_out, P);
    const int nwords = csr_words(V, h, w);
    hipLaunchKernelGGL(csr_count_kernel, dim3(R), dim3(256), nwords * 4, st, rect, match, pos2s, bits_ws, row_count, h, w, V, topk, nwords);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(R), dim3(256), 0, st, bits_ws, row_count, pos2s, rect, row_ptr, col_idx, nnz_out, R, nwords,
                       V * h * w, V, col_cap);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// S-path helper: positions touched by RoIAlign taps (own rects expanded by `expand_stride` cells) -> compact list.
extern "C" int mv2d_roi_positions(const float* rois, const unsigned char* pad_mask, unsigned char* roi_mask, int* rect, int* pos2s,
                                  int* s2pos, int* S_out, int R, int V, int h, int w, float stride, float expand_stride, void* stream) {
    MV2D_CHECK_ARG(rois && pad_mask && roi_mask && rect && pos2s && s2pos && S_out && R > 0, "mv2d_roi_positions: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(csr_mark_kernel, dim3(R), dim3(64), 0, st, rois, rect, roi_mask, h, w, stride, expand_stride);
    hipLaunchKernelGGL(csr_scan_positions_kernel, dim3(cdiv(V * h * w, SCAN_SEG)), dim3(1024), 0, st, roi_mask, pad_mask, pos2s, s2pos, S_out, V * h * w);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_csr_from_corr(const int* match, int* row_ptr, int* col_idx, int* nnz_out, int R, int V, int topk, void* stream) {
    MV2D_CHECK_ARG(match && row_ptr && col_idx && nnz_out && R > 0 && V * topk >= 0 && V * topk <= 4096, "mv2d_csr_from_corr: bad args (V * topk <= 4096)");
    hipLaunchKernelGGL(csr_from_corr_kernel, dim3(cdiv(R, CFC_ROWS)), dim3(1024), 0, (hipStream_t)stream, match, row_ptr, col_idx, nnz_out, R, V * topk);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// mv2d_roi_positions + mv2d_csr_from_corr (S path) in TWO launches instead of three: the mark kernel, then the position scan and the CSR side by side
extern "C" int mv2d_roi_positions_csr(const float* rois, const unsigned char* pad_mask, unsigned char* roi_mask, int* rect, int* pos2s, int* s2pos,
                                      int* S_out, int R, int V, int h, int w, float stride, float expand_stride, const int* match, int* row_ptr,
                                      int* col_idx, int* nnz_out, int Vg, int topk, const int* grp_start, int n_samples, int* order, int* order_flags, void* stream) {
    MV2D_CHECK_ARG(!order || (grp_start && n_samples >= 1), "mv2d_roi_positions_csr: the block order needs the sample row ranges");
    MV2D_CHECK_ARG(rois && pad_mask && roi_mask && rect && pos2s && s2pos && S_out && R > 0, "mv2d_roi_positions_csr: bad args");
    MV2D_CHECK_ARG(match && row_ptr && col_idx && nnz_out && Vg * topk >= 0 && Vg * topk <= 4096, "mv2d_roi_positions_csr: bad CSR args (views per sample * topk <= 4096)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(csr_mark_kernel, dim3(R), dim3(64), 0, st, rois, rect, roi_mask, h, w, stride, expand_stride);
    const int nscan = cdiv(V * h * w, SCAN_SEG), ncsr = cdiv(R, CFC_ROWS);
    const int ord_chunks = cdiv(R < SORD_MAX ? R : SORD_MAX, SORD_CHUNK), nord = order ? (n_samples + 1) * ord_chunks : 0;
    hipLaunchKernelGGL(scan_and_csr_kernel, dim3(nscan + ncsr + nord), dim3(1024), 0, st, roi_mask, pad_mask, pos2s, s2pos, S_out, V * h * w, nscan,
                       match, row_ptr, col_idx, nnz_out, R, Vg * topk, ncsr, grp_start, n_samples, ord_chunks, order, order_flags);

    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_pe_frustum_f32(const int* s2pos, const int* S_dev, int S_max, const double* img2lidar, const double* coords_w, const double* coords_h,
                                   const double* coords_d, float* out, int V, int h, int w, int depth_num, const double* position_range, void* stream) {
    MV2D_CHECK_ARG(s2pos && S_dev && img2lidar && coords_w && coords_h && coords_d && out && position_range, "mv2d_pe_frustum_f32: null pointer");
    MV2D_CHECK_ARG(depth_num > 0 && depth_num <= 256 && V > 0 && h > 0 && w > 0, "mv2d_pe_frustum_f32: bad sizes");
    if (S_max == 0) return MV2D_OK;
    // the logarithm table travels as a kernel argument (round 6; it was a __device__ symbol filled once per PROCESS: a second GPU of the process read
    // zeros, and a first call under graph capture broke the capture -- ADVICE r5); built once, thread-safe
    static const LogTab lt = [] {
        LogTab t;
        for (int i = 0; i < 128; ++i) {
            const double c = 1.0 + (i + 0.5) / 128.0;
            t.t[2 * i] = 1.0 / c;
            t.t[2 * i + 1] = log(c);
        }
        return t;
    }();
    const int blocks = cdiv(S_max, 4) < 4096 ? cdiv(S_max, 4) : 4096;
    hipLaunchKernelGGL(pe_frustum_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s2pos, S_dev, img2lidar, coords_w, coords_h, coords_d, out, h, w,
                       depth_num, position_range[0], position_range[1], position_range[2], 1.0 / (position_range[3] - position_range[0]),
                       1.0 / (position_range[4] - position_range[1]), 1.0 / (position_range[5] - position_range[2]), lt);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_pe_inputs(const int* s2pos, const int* S_dev, int S_max, const float* featcl, const double* img2lidar,
                              const double* coords_w, const double* coords_h, const double* coords_d, const float* embeds,
                              const float* dim_t, void* A_frustum, void* A_sine, void* Xf_k16, float* Xf_f32, float* A_frustum_f32,
                              float* A_sine_f32, int V, int h, int w, int depth_num, const double* position_range, void* stream) {
    MV2D_CHECK_ARG(s2pos && S_dev && featcl && img2lidar && coords_w && coords_h && coords_d && embeds && dim_t && A_frustum &&
                       Xf_k16 && position_range, "mv2d_pe_inputs: null pointer");
    MV2D_CHECK_ARG(A_sine || !A_sine_f32, "mv2d_pe_inputs: the exact rows need A_sine");
    MV2D_CHECK_ARG(depth_num <= 256 && (depth_num % 8) == 0, "mv2d_pe_inputs: depth_num must be a multiple of 8, <= 256");
    if (S_max == 0) return MV2D_OK;
    MV2D_CHECK_ARG(A_frustum_f32 || !A_sine_f32, "mv2d_pe_inputs: fp32 sine rows only together with the fp32 frustum rows");
#define MV2D_PEI(EX) hipLaunchKernelGGL(pe_inputs_kernel<EX>, dim3(cdiv(S_max, 4)), dim3(256), 0, (hipStream_t)stream, s2pos, S_dev, featcl, img2lidar, coords_w, \
                       coords_h, coords_d, embeds, dim_t, (unsigned short*)A_frustum, (unsigned short*)A_sine,                                      \
                       (unsigned short*)Xf_k16, Xf_f32, A_frustum_f32, A_sine_f32, h, w, V * h * w, depth_num, position_range[0], position_range[1], \
                       position_range[2], position_range[3] - position_range[0], position_range[4] - position_range[1],                               \
                       position_range[5] - position_range[2])
    if (A_frustum_f32) MV2D_PEI(true); else MV2D_PEI(false);
#undef MV2D_PEI
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_result_pack(const float* boxes, const float* scores, const long long* labels, const int* count, float score_thr, int max_num,
                                float* out_boxes, float* out_scores, long long* out_labels, int* out_count, int n_samples, int in_stride,
                                void* stream) {
    MV2D_CHECK_ARG(boxes && scores && labels && count && out_boxes && out_scores && out_labels && out_count, "mv2d_result_pack: null pointer");
    MV2D_CHECK_ARG(max_num >= 1 && max_num <= 1024 && n_samples >= 1, "mv2d_result_pack: max_num must be in [1, 1024], n_samples >= 1");
    hipLaunchKernelGGL(result_pack_kernel, dim3(n_samples), dim3(1024), 0, (hipStream_t)stream, boxes, scores, labels, count, score_thr, max_num,
                       out_boxes, out_scores, out_labels, out_count, in_stride);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_decode_topk(const float* cls, const float* reg, int R, int num_classes, int max_num, const float* post_center_range,
                                float* boxes, float* scores, long long* labels, long long* bbox_index, int* count_out,
                                long long* topk_index_dbg, const int* grp_start, int n_grp, int max_grp_rows, float* payload, void* stream) {
    MV2D_CHECK_ARG(cls && reg && post_center_range && boxes && scores && labels && bbox_index && count_out, "mv2d_decode_topk: null pointer");
    MV2D_CHECK_ARG(max_num >= 1 && max_num <= 1024, "mv2d_decode_topk: max_num must be in [1, 1024]");
    MV2D_CHECK_ARG(!grp_start || (n_grp >= 1 && max_grp_rows >= 1 && max_grp_rows <= R), "mv2d_decode_topk: bad sample list");
    const int n = (grp_start ? max_grp_rows : R) * num_classes;
    MV2D_CHECK_ARG(n > 0 && n <= 16384, "mv2d_decode_topk: rows (of one sample) * num_classes must be in [1, 16384]");
    int npow2 = 1024;
    while (npow2 < n) npow2 <<= 1;
    const size_t lds = (size_t)(npow2 + 1024) * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)decode_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (16384 + 1024) * 8);
        attr_set = true;
    }
    hipLaunchKernelGGL(decode_topk_kernel, dim3(grp_start ? n_grp : 1), dim3(1024), lds, (hipStream_t)stream, cls, reg, R, num_classes, max_num, npow2,
                       post_center_range[0], post_center_range[1], post_center_range[2], post_center_range[3], post_center_range[4],
                       post_center_range[5], boxes, scores, labels, bbox_index, count_out, topk_index_dbg, grp_start, payload);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_pack_detections(const float* boxes, const float* scores, const long long* labels, const int* count, float* out, int n_samples,
                                    int max_num, int in_stride, void* stream) {
    MV2D_CHECK_ARG(boxes && scores && labels && count && out && n_samples >= 1 && max_num >= 1 && in_stride >= max_num, "mv2d_pack_detections: bad args");
    hipLaunchKernelGGL(pack_detections_kernel, dim3(n_samples), dim3(256), 0, (hipStream_t)stream, boxes, scores, labels, count, out, max_num, in_stride);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
