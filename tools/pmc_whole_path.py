"""Whole-path HBM traffic per sample from the per-kernel PMC tables of tools/pmc_bench.sh:
   python tools/pmc_whole_path.py gpurun_out/<dir> <samples_per_launch>
Sum over the kernels of one frame of (FETCH_SIZE x 2 + WRITE_SIZE) [KiB -> bytes] x launches per frame.  The run behind the tables also holds
warm-up frames and the decoder-only timing leg, so the per-launch AVERAGE of every kernel is multiplied by its multiplicity inside one frame
(once-per-frame kernels: 1; per decoder layer: 6; out projections: 12) instead of dividing totals by a frame count."""
import re, sys

PER_FRAME = {'xattn_fused_kernel': 6, 'xattn_tile_kernel': 6, 'xattn_qmap_kernel': 6, 'xattn_ctxmap_kernel': 6, 'ffn_x3_kernel': 6, 'ffn_out_fused_x3_kernel': 6, 'attn_out_fused_x3_kernel': 12,
             'attn_out_qmap': 6, 'attn_out_zmap': 6, 'self_attn_x3_kernel': 6, 'self_attn_kernel': 6}
SKIP = ('spin_kernel', 'pack_wfrag', 'split_bf16x2', 'split_q16x2', 'f32_to_key16', 'gemm_bf16', 'split3_rows', 'at6native', 'rocclr', 'elementwise')


def table(path):
    rows = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and f[2] in ('FETCH_SIZE', 'WRITE_SIZE'):
            rows.setdefault((f[0], f[1]), []).append((int(f[3]), float(f[4])))
    return rows


def main():
    d, B = sys.argv[1], int(sys.argv[2])
    fe, wr = table(d + '/FETCH_SIZE.txt'), table(d + '/WRITE_SIZE.txt')
    frames = max(c for (k, _), v in fe.items() for c, _ in v if 'decode_topk' in k)
    out, tot = [], 0.0
    for key, v in fe.items():
        k, grid = key
        if any(s in k for s in SKIP):
            continue
        calls = sum(c for c, _ in v)
        if calls < frames:
            continue                                                   # set-up kernels (once per weights / rig)
        f = sum(c * a for c, a in v) / calls * 2048.0
        w = sum(c * a for c, a in wr.get(key, [(1, 0.0)])) / max(sum(c for c, _ in wr.get(key, [(1, 0.0)])), 1) * 1024.0
        mult = next((m for n, m in PER_FRAME.items() if n in k), 1)
        b = (f + w) * mult
        tot += b
        out.append((b, mult, re.sub(r'_ZN12_GLOBAL__N_1\d+', '', k)[:44], grid, f, w))
    out.sort(reverse=True)
    print(f'{d}: {tot / 1e6:.0f} MB per {B}-sample frame = {tot / B / 1e6:.1f} MB per sample (FETCH_SIZE x 2 + WRITE_SIZE, all kernels of a frame)')
    for b, m, k, g, f, w in out:
        print(f'  {b / 1e6:8.1f} MB  = {m:2d} x (fetch {f / 1e6:7.1f} + write {w / 1e6:7.1f})  {k} [{g}]')


if __name__ == '__main__':
    main()
