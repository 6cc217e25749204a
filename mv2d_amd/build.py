"""Build libmv2d_hip.so (hand-written gfx950 HIP kernels + C-ABI) in-tree with hipcc.

    python -m mv2d_amd.build            # compiles mv2d_amd/csrc/*.hip -> mv2d_amd/lib/libmv2d_hip.so

hipcc cross-compiles for gfx950 without a GPU present.  The .so is git-ignored but travels to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libmv2d_hip.so')
OBJDIR = os.path.join(LIBDIR, 'obj')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-value']


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (set HIPCC=...)')


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def is_fresh():
    stamp = LIB + '.sha256'
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _digest()


def build(force=False, verbose=True):
    if not force and is_fresh():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    keep = {os.path.basename(f)[:-4] + '.o' for f in srcs}
    for f in os.listdir(OBJDIR):                      # objects of sources that no longer exist (tools/build_variant.sh links every object here)
        if f.endswith('.o') and f not in keep:
            os.remove(os.path.join(OBJDIR, f))

    def cc(src):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + '.o')
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed: %s\n%s' % (' '.join(cmd), r.stderr))
    with open(LIB + '.sha256', 'w') as fh:
        fh.write(_digest())
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


def build_variant(name, defs, verbose=True):
    """A whole-library A/B build with extra -D flags (e.g. -DMV2D_Q16_BF16: the round-4 query-side format) ->
    mv2d_amd/lib/variants/lib<name>.so; load it with MV2D_HIP_LIB=... (tools/ab_lib.sh)."""
    hipcc = _hipcc()
    vdir = os.path.join(LIBDIR, 'variants')
    odir = os.path.join(vdir, 'obj_' + name)
    os.makedirs(odir, exist_ok=True)

    def cc(src):
        obj = os.path.join(odir, os.path.basename(src)[:-4] + '.o')
        r = subprocess.run([hipcc] + FLAGS + list(defs) + ['-c', src, '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, _sources()))
    out = os.path.join(vdir, 'lib%s.so' % name)
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr)
    shutil.rmtree(odir)
    if verbose:
        print('built', out, file=sys.stderr)
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:
        i = sys.argv.index('--variant')
        build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith('-D')])
    else:
        build(force='--force' in sys.argv)
