// T path: launch order of the per-query blocks of the tile cross attention (gfx950 / CDNA4, wave64).
//
// On the masked-map path (MV2DTHead, RH/mv2d_t_head.py:79-109; PETRMultiheadAttention with a boolean attn_mask,
// MU/petr_transformer.py:501-508) every key row is read by 2.9 (cfg3_t) to 6.2 (cfg5_t) queries.  Queries that share keys are neighbours
// once the queries of a sample are ORDERED BY THEIR SMALLEST KEY INDEX (the key list is in (view, y, x) order: a query and the queries of
// the RoIs it is epipolar-matched with start at the same cells).  xattn_tile_kernel launched in that order (its `order` argument) runs
// blocks with overlapping key sets side by side on one XCD, so the repeats are served by that XCD's L2: cfg3_t 54.8 -> 47.6 us,
// cfg5_t 60.2 -> 51.7 us per layer, bitwise the same results (round 3).
// (Round 3 also built a shared-key-tile kernel on this order -- 16 queries per workgroup, the union of their key lists streamed once
//  through an LDS-DMA ring, 16-bit pair masks.  It read 1.40 x the distinct rows instead of 2.99 x and was 3 x SLOWER: a union tile is only
//  ~35 % allowed pairs for a given pair of queries, the masked MFMA / softmax work tripled.  Retired in round 4; LOG.md section 8.)
#include "common.h"

namespace {

constexpr int GRP_MAX = 4096;                // queries per sample the ordering kernel ranks in LDS
constexpr int ORD_CHUNK = 128;               // queries ranked per block (8 threads per query)

// groups: sample b = rows [grp_start[b], grp_start[b+1]), b < n_grp; the bucket-padding rows [grp_start[n_grp], R) form one more group.
// perm[slot] = query of sorted slot `slot` (slots of a group = its row range).
// grid (groups, chunks of ORD_CHUNK queries): every block holds the group's keys in LDS and ranks its chunk, 8 threads per query
// (one block per group took 39 us for the ~950-query samples of cfg5_t)
__global__ __launch_bounds__(1024) void query_order_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const int* __restrict__ grp_start,
                                                           int n_grp, int R, int* __restrict__ perm, int* __restrict__ flags, int stride) {
    __shared__ int key[GRP_MAX];
    const int g = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int lo = g < n_grp ? grp_start[g] : grp_start[n_grp];
    const int hi = g < n_grp ? grp_start[g + 1] : R;
    const int n = hi - lo;
    if (n > GRP_MAX) {                                         // too many queries in one sample for the LDS ranking: keep the natural order
        if (chunk == 0) {
            if (tid == 0) flags[0] = 1;
            for (int i = tid; i < n; i += 1024) perm[lo + i] = lo + i;
        }
        return;
    }
    if (chunk * ORD_CHUNK >= n) return;
    for (int i = tid; i < n; i += 1024) {
        const int b = row_ptr[lo + i], e = row_ptr[lo + i + 1];
        int k = e > b ? col_idx[b] : 0x7fffffff;               // T path: the CSR rows are ascending, the first entry is the smallest key
        // S path (stride = 49): a row lists whole RoIs (own RoI first, then the matched ones in view order): the smallest first cell
        if (stride > 0) for (int q = b + stride; q < e; q += stride) k = min(k, col_idx[q]);
        key[i] = k;
    }
    __syncthreads();
    const int i = chunk * ORD_CHUNK + (tid >> 3), sub = tid & 7;
    int rank = 0;
    if (i < n) {
        const int ki = key[i];
        for (int j = sub; j < n; j += 8) { const int kj = key[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    rank += __shfl_xor(rank, 4);
    if (i < n && sub == 0) perm[lo + rank] = lo + i;
}

}  // namespace

// perm [R] = the queries of every sample sorted by their smallest key (bucket-padding rows behind the last sample keep their places as a
// group of their own); stride = 0: the first entry of a CSR row is its smallest key (T path: ascending rows); stride > 0: the minimum over
// every stride-th entry (S path, stride 49: the first cell of every RoI a row lists -- queries of RoIs that are matched with each other
// then sit side by side, round 4: the overlapping-rig workload re-fetched 1.86 x its distinct rows from HBM in the natural order); flags [>= 1] int32, zeroed by the caller: flags[0] != 0 afterwards = a sample had more than 4096 queries and kept its
// natural order (any order is correct).
extern "C" int mv2d_xattn_query_order(const int* row_ptr, const int* col_idx, const int* grp_start, int n_samples, int R, int* perm, int* flags, int stride,
                                      void* stream) {
    MV2D_CHECK_ARG(row_ptr && col_idx && grp_start && perm && flags && R > 0 && n_samples >= 1 && stride >= 0, "mv2d_xattn_query_order: bad args");
    const int n = R < GRP_MAX ? R : GRP_MAX;
    hipLaunchKernelGGL(query_order_kernel, dim3(n_samples + 1, (n + ORD_CHUNK - 1) / ORD_CHUNK), dim3(1024), 0, (hipStream_t)stream, row_ptr, col_idx,
                       grp_start, n_samples, R, perm, flags, stride);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
