"""Register / LDS / scratch use of every kernel of libnvdr_hip.so: each translation unit is compiled to gfx950 assembly with the flags
of the in-tree build and the .amdhsa_kernel blocks are read.  usage: python tools/kernel_resources.py [out.md]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrecmc_amd import _build  # noqa: E402

rows = []
with tempfile.TemporaryDirectory() as tmp:
    for src in _build.SOURCES:
        out = os.path.join(tmp, src + '.s')
        cmd = [_build._hipcc()] + _build.FLAGS + ['-S', '--cuda-device-only', '-o', out, os.path.join(_build.CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            raise SystemExit(1)
        s = open(out).read()
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
            name, body = m.group(1), m.group(2)
            if 'rocprim' in name:
                continue
            g = lambda k: int(re.search(k + r'\s+(\S+)', body).group(1))
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r'\(.*', '', dem).replace('void ', '')
            spill = re.search(r'; codeLenInByte.*?', s)
            # the per-function comment block carries the spill count
            fm = re.search(re.escape(name) + r':.*?; ScratchSize: (\d+).*?', s, re.S)
            sp = re.search(r'\.vgpr_spill_count:\s+(\d+)', s[s.find('.name:           ' + name):s.find('.name:           ' + name) + 4000]) if ('.name:           ' + name) in s else None
            rows.append((src.replace('.hip', ''), dem, g(r'\.amdhsa_next_free_vgpr'), int(sp.group(1)) if sp else 0, g(r'\.amdhsa_next_free_sgpr'),
                         g(r'\.amdhsa_group_segment_fixed_size'), g(r'\.amdhsa_private_segment_fixed_size')))
rows.sort(key=lambda r: (r[0], -r[2], r[1]))
lines = ['# Register / LDS / scratch use of the kernels of libnvdr_hip.so',
         '',
         '`python tools/kernel_resources.py`: every translation unit compiled with the flags of the in-tree build (`-O3 --offload-arch=gfx950 -ffp-contract=off`),',
         'code-object metadata (rocPRIM\'s sort kernels left out).  VGPRs per lane decide the waves per SIMD (512 / VGPRs, allocated in steps of 8);',
         'LDS = static bytes per workgroup (the traversal, filter, gather and G-buffer kernels add dynamic LDS at launch).',
         '',
         '| file | kernel | VGPRs | spilled | SGPRs | static LDS B | scratch B per lane |', '|---|---|---|---|---|---|---|']
for r in rows:
    lines.append('| %s | `%s` | %d | %d | %d | %d | %d |' % r)
text = '\n'.join(lines) + '\n'
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write(text)
print(text)
