"""Fold the per-kernel FETCH_SIZE / WRITE_SIZE tables of tools/pmc_bench.sh into profiles/pmc_traffic.json:
   python tools/pmc_to_json.py gpurun_out/<dir> <workload>@<samples_per_launch> "<provenance note>"
Per-launch HBM bytes of the kernels bench.py's roofline objects describe (average over the run's calls; FETCH_SIZE is in KiB and is
doubled on gfx950, MI355X_MICROARCH.md section HBM)."""
import json, os, sys

KERNELS = {'pe_fused': ('pe_x3_kernel', 'pe_tab_kernel', 'pe_fused_kernel'), 'xattn_fused': ('xattn_fused_kernel',), 'pe_frustum': ('pe_frustum_f32_kernel',), 'nchw_to_nhwc': ('nchw_to_nhwc',), 'pe_inputs': ('pe_inputs_kernel',), 'qg_conv_gemm': ('roi_conv_pool_kernel',), 'xattn_tile': ('xattn_tile_kernel',),
           'roi_align': ('roi_align_kernel',), 'self_attn': ('self_attn_x3_kernel', 'self_attn_kernel'), 'ffn': ('ffn_x3_kernel',)}


def table(path):
    rows = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and f[2] in ('FETCH_SIZE', 'WRITE_SIZE'):
            rows.setdefault(f[0], []).append((int(f[3]), float(f[4])))          # calls, average KiB
    return rows


def pick(rows, names):
    best = None
    for k, v in rows.items():
        if any(n in k for n in names):
            calls = sum(c for c, _ in v)
            avg = sum(c * a for c, a in v) / calls
            if best is None or calls * avg > best[0] * best[1]:
                best = (calls, avg, k)
    return best


def main():
    d, key, note = sys.argv[1], sys.argv[2], sys.argv[3]
    fe, wr = table(os.path.join(d, 'FETCH_SIZE.txt')), table(os.path.join(d, 'WRITE_SIZE.txt'))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'profiles', 'pmc_traffic.json')
    js = json.load(open(path))
    out = {}
    for name, kn in KERNELS.items():
        f, w = pick(fe, kn), pick(wr, kn)
        if f and w:
            out[name] = dict(kernel=f[2][:60], fetch_bytes=int(f[1] * 1024 * 2), write_bytes=int(w[1] * 1024), calls=f[0])
    whole_f = sum(c * a for v in fe.values() for c, a in v) * 1024 * 2
    whole_w = sum(c * a for v in wr.values() for c, a in v) * 1024
    out['_whole_run'] = dict(fetch_bytes=int(whole_f), write_bytes=int(whole_w), note='all kernels of the profiled run (warm-up + 3 steps + stage timing + set-up)')
    out['_provenance'] = note
    # stamp: digest of the kernel sources the counters were measured on (mv2d_amd/build.py writes it next to the library); bench.py flags the
    # entry as stale when the library it runs has another digest
    try:
        out['_csrc_sha256'] = open(os.path.join(root, 'mv2d_amd', 'lib', 'libmv2d_hip.so.sha256')).read().strip()
    except OSError:
        out['_csrc_sha256'] = None
    js[key] = out
    json.dump(js, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
