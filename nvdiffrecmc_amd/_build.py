"""In-tree build of libnvdr_hip.so (explicit hipcc for gfx950; no JIT cache, no cmake).

The library is a plain C-ABI shared object (include/nvdr_hip.h) with no libtorch linkage.  It
links the HIP runtime by SONAME (libamdhip64.so.7); inside a Python process that has imported
torch first, the loader resolves that to the runtime torch already mapped, so device pointers
and stream handles coming from torch are valid in it.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nvdiffrecmc_amd', 'csrc')
BUILD = os.path.join(CSRC, 'build')
LIB = os.path.join(BUILD, 'libnvdr_hip.so')
ARCH = 'gfx950'

SOURCES = [
    'core.hip',
    'bvh.hip',
    'env_shade.hip',
    'denoise.hip',
    'renderutils.hip',
    'light.hip',
    'gbuffer.hip',
    'mesh.hip',
    'optim.hip',
    'exchange.hip',
]

# -ffp-contract=off: the sampling math must round exactly like the CPU oracle
# (see include/nvdr_detmath.h); fused operations are written as explicit fmaf.
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-Wno-unused-result', '-Wno-unused-value'] + os.environ.get('NVDR_EXTRA_FLAGS', '').split()


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libnvdr_hip.so')
    return exe


def _deps_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, 'include')):
        for fn in os.listdir(d):
            if fn.endswith(('.h', '.hpp')):
                m = max(m, os.path.getmtime(os.path.join(d, fn)))
    return m


def _compile(src, force, hdr_mtime, verbose):
    obj = os.path.join(BUILD, os.path.splitext(src)[0] + '.o')
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(sp)
            and os.path.getmtime(obj) > hdr_mtime):
        return obj, False
    cmd = [_hipcc()] + FLAGS + ['-c', sp, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj, True


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libnvdr_hip.so.  Returns its path."""
    os.makedirs(BUILD, exist_ok=True)
    hdr_mtime = _deps_mtime()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    missing = [s for s in SOURCES if s not in srcs]
    if missing:
        raise RuntimeError('missing HIP sources: %s' % missing)
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_mtime, verbose), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-ldl', '-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
