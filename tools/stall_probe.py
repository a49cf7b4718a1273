"""What the traversal kernel's wavefronts wait for, by hardware counter: rocprofv3 --pmc passes (kernel trace only) over a few env-shade passes of
tools/ab_inproc.py, one pass per small group of counters; the names are taken from `rocprofv3 --list-avail` of the box, so an unknown counter
drops out instead of failing its pass.
    PROBE_VIEWS=8 python tools/stall_probe.py out.md [kernel substring, default env_trace_kernel]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_parts import _pmc_read  # noqa: E402

GROUPS = [
    ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_MISC'],
    ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_WAIT_IFETCH', 'SQ_IFETCH', 'SQ_WAVES', 'SQ_ACTIVE_INST_FLAT'],
    ['SQ_WAVE_CYCLES', 'SQ_INST_CYCLES_VMEM_RD', 'SQ_INST_CYCLES_VMEM_WR', 'SQ_INST_CYCLES_SALU', 'SQ_INST_CYCLES_SMEM', 'SQ_INSTS_BRANCH', 'SQ_INSTS_SENDMSG', 'SQ_INSTS_VSKIPPED'],
    ['SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_ADDR_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_UNALIGNED_STALL', 'SQ_LDS_MEM_VIOLATIONS', 'SQ_INSTS_LDS', 'SQ_LDS_ATOMIC_RETURN'],
    ['SQ_WAVE_CYCLES', 'SQC_ICACHE_REQ', 'SQC_ICACHE_HITS', 'SQC_ICACHE_MISSES', 'SQC_DCACHE_REQ', 'SQC_DCACHE_HITS', 'SQC_DCACHE_MISSES', 'SQC_TC_REQ'],
    ['GRBM_GUI_ACTIVE', 'TA_TA_BUSY_sum', 'TA_BUSY_avr', 'TA_FLAT_READ_WAVEFRONTS_sum', 'TA_BUFFER_WAVEFRONTS_sum'],
    ['GRBM_GUI_ACTIVE', 'TA_ADDR_STALLED_BY_TC_CYCLES_sum', 'TA_DATA_STALLED_BY_TC_CYCLES_sum', 'TA_ADDR_STALLED_BY_TD_CYCLES_sum', 'TA_FLAT_WAVEFRONTS_sum'],
    ['GRBM_GUI_ACTIVE', 'TD_TD_BUSY_sum', 'TD_TC_STALL_sum', 'TD_LOAD_WAVEFRONT_sum', 'TD_COALESCABLE_WAVEFRONT_sum'],
    ['GRBM_GUI_ACTIVE', 'TCP_GATE_EN1_sum', 'TCP_GATE_EN2_sum', 'TCP_TD_TCP_STALL_CYCLES_sum', 'TCP_TCR_TCP_STALL_CYCLES_sum'],
    ['GRBM_GUI_ACTIVE', 'TCP_PENDING_STALL_CYCLES_sum', 'TCP_READ_TAGCONFLICT_STALL_CYCLES_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum'],
    ['GRBM_GUI_ACTIVE', 'TCP_TA_TCP_STATE_READ_sum', 'TCP_TOTAL_ACCESSES_sum', 'TCP_TOTAL_READ_sum', 'TCP_TCC_READ_REQ_LATENCY_sum'],
]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else '/dev/stdout'
    needle = sys.argv[2] if len(sys.argv) > 2 else 'env_trace_kernel<false>'
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    env = dict(os.environ, TMPDIR='/tmp', NVDR_TUNING='1', AB_ONLY='current', AB_ITERS='3')
    avail = subprocess.run([exe, '--list-avail'], cwd='/tmp', env=env, capture_output=True, text=True)
    names = set(re.findall(r'[A-Z][A-Za-z0-9_]{3,}', avail.stdout + avail.stderr))
    lines = ['# Hardware counters of `%s` (rocprofv3 --pmc, one pass per group; PROBE_VIEWS=%s PROBE_RES=%s PROBE_SUBDIV=%s PROBE_MESH=%s)' %
             (needle, os.environ.get('PROBE_VIEWS', '8'), os.environ.get('PROBE_RES', '512'), os.environ.get('PROBE_SUBDIV', '0'), os.environ.get('PROBE_MESH', 'bob')),
             '', 'per-launch sums over the non-empty dispatches; counters the box does not list are left out (%d names listed)' % len(names), '',
             '| pass | counter | per launch | launches |', '|---|---|---|---|']
    for gi, group in enumerate(GROUPS):
        g = [c for c in group if c in names] if names else group
        missing = [c for c in group if c not in g]
        if not g:
            lines.append('| %d | (none of %s listed) | | |' % (gi, ' '.join(group)))
            continue
        d = tempfile.mkdtemp(prefix='nvdr_stall%d_' % gi, dir='/tmp')
        cmd = [exe, '--kernel-trace', '--pmc'] + g + ['-d', d, '-o', 'r', '--', sys.executable, os.path.join(ROOT, 'tools', 'ab_inproc.py'), '1']
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=400)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith('_results.db')]
            if r.returncode != 0 or not dbs:
                lines.append('| %d | FAILED rc=%d (%s): %s | | |' % (gi, r.returncode, ' '.join(g), (r.stderr or r.stdout)[-200:].replace('\n', ' ')))
                continue
            ctrs, disp = _pmc_read(dbs[0], g[0])
            for kname, c in ctrs.items():
                if needle in kname:
                    for ctr in g:
                        if ctr in c:
                            lines.append('| %d | %s | %.6g | %d |' % (gi, ctr, c[ctr], disp.get(kname, 0)))
            if missing:
                lines.append('| %d | (not listed: %s) | | |' % (gi, ' '.join(missing)))
        except subprocess.TimeoutExpired:
            lines.append('| %d | TIMEOUT (%s) | | |' % (gi, ' '.join(g)))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    open(out_path, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
