"""The drop-in boundary: libnvdr_hip.so loads on a machine without a GPU and exports every symbol that
include/nvdr_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest
import torch

from nvdiffrecmc_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'nvdr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nvdr_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_all_exported():
    assert os.path.exists(_build.LIB), 'build the HIP library first (__graft_entry__.build())'
    lib = ctypes.CDLL(_build.LIB)
    names = _declared()
    assert len(names) >= 39
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared()
    lib = _lib.load()
    assert lib.nvdr_version() >= 100
    assert lib.nvdr_image_loss_num_partials(1, 512, 512) == 1024
    assert lib.nvdr_image_loss_num_partials(1, 8, 8) == 1


def test_struct_layouts_match_the_header():
    # nvdr_tensor: pointer + 4 sizes + 4 strides; the env-shade block: 12 tensors, 5 scalars (+pad), 2 ptrs, 2 tensors, 9 ptrs / words, snapshot ptr + advance + phase
    assert ctypes.sizeof(_lib.NvdrTensor) == 8 + 4 * 8 + 4 * 8
    assert ctypes.sizeof(_lib.NvdrEnvShadeArgs) == 12 * 72 + 24 + 2 * 8 + 2 * 72 + 9 * 8 + 16
    from oracle import oracle as orc
    assert ctypes.sizeof(orc.EnvShadeArgs) == ctypes.sizeof(_lib.NvdrEnvShadeArgs)
    # the geometry / material gradient route (round 4): sizes as a C compiler lays the header's structs out
    assert ctypes.sizeof(_lib.NvdrMeshArgs) == 8 * 8
    assert ctypes.sizeof(_lib.NvdrInterpolateBwdArgs) == 8 + 4 * 4 + 4 * 8 + 2 * 8 + 8 + 4 * 8 + 3 * 8      # rast, n h w (+pad), 4 ptrs, 2 counts, cam, 4 + 3 ptrs
    assert ctypes.sizeof(_lib.NvdrTextureArgs) == 4 + 4 * 4 + 4 + 4 * 8 + 2 * 8 + 8 + 3 * 4 * 8 + 8        # ... + accumulate (+pad)


def test_new_struct_layouts_against_the_c_compiler(tmp_path):
    """sizeof / offsetof of the round-4 argument blocks, asked of gcc on include/nvdr_hip.h itself."""
    import subprocess
    src = tmp_path / 'lay.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nvdr_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(nvdr_mesh_args), sizeof(nvdr_interpolate_bwd_args), sizeof(nvdr_texture_args), offsetof(nvdr_interpolate_bwd_args, cam),'
                   'offsetof(nvdr_interpolate_bwd_args, v_tng_grad), offsetof(nvdr_texture_args, texc), offsetof(nvdr_texture_args, dtex));'
                   'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(nvdr_adam_tensor), offsetof(nvdr_adam_tensor, lr_scale), offsetof(nvdr_adam_tensor, active),'
                   'sizeof(nvdr_env_shade_args), offsetof(nvdr_env_shade_args, rnd_seed_snapshot), offsetof(nvdr_texture_args, accumulate));'
                   'printf("%zu\\n", offsetof(nvdr_env_shade_args, phase));return 0;}\n')
    exe = tmp_path / 'lay'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    A, T, D, E = _lib.NvdrInterpolateBwdArgs, _lib.NvdrTextureArgs, _lib.NvdrAdamTensor, _lib.NvdrEnvShadeArgs
    assert got == [ctypes.sizeof(_lib.NvdrMeshArgs), ctypes.sizeof(A), ctypes.sizeof(T), A.cam.offset, A.v_tng_grad.offset, T.texc.offset, T.dtex.offset,
                   ctypes.sizeof(D), D.lr_scale.offset, D.active.offset, ctypes.sizeof(E), E.rnd_seed_snapshot.offset, T.accumulate.offset, E.phase.offset]


def test_no_cpu_fallback_errors_are_loud(monkeypatch):
    import nvdiffrecmc_amd.renderutils as ru
    import nvdiffrecmc_amd.optixutils as ou
    x = torch.rand(1, 4, 4, 3)
    with pytest.raises(RuntimeError, match='GPU'):
        ru.lambert(x, x)
    with pytest.raises(RuntimeError, match='GPU'):
        ru.image_loss(x, x)
    with pytest.raises(RuntimeError, match='GPU'):
        ou.bilateral_denoiser(x, x, x[..., :2], 2.0)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ou.OptiXContext()
    # a missing library is an error, never a silent fallback
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_build, 'LIB', '/nonexistent/libnvdr_hip.so')
    with pytest.raises(RuntimeError, match='not built'):
        _lib.load()


def test_roctx_ranges_load_the_marker_library_on_demand():
    """NVDR_ROCTX=1: the first entry point that opens a range loads the roctx library (nothing is linked against the profiler);
    without the variable the library is not touched.  The call itself fails on its NULL argument -- no GPU work here."""
    import subprocess
    import sys
    prog = ('import ctypes, os\n'
            'lib = ctypes.CDLL(%r)\n'
            'lib.nvdr_last_error.restype = ctypes.c_char_p\n'
            'r = lib.nvdr_adam_step(None, 0, ctypes.c_double(1e-3), ctypes.c_double(.9), ctypes.c_double(.999), ctypes.c_double(1e-8), None, None)\n'
            'assert r == -1 and b"NULL" in lib.nvdr_last_error(), (r, lib.nvdr_last_error())\n'
            'print("roctx" in open("/proc/self/maps").read())\n' % _build.LIB)
    for flag, want in (('1', 'True'), ('0', 'False'), (None, 'False')):
        env = {k: v for k, v in os.environ.items() if k != 'NVDR_ROCTX'}
        if flag is not None:
            env['NVDR_ROCTX'] = flag
        out = subprocess.run([sys.executable, '-c', prog], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want, (flag, out.stdout, out.stderr)
