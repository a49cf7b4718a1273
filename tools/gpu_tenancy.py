"""Who else is on this GPU?  Prints, for every DRM card the container can see, the device-wide busy percentages and VRAM / GTT usage
(sysfs), the KFD process list, and this process's own share.  Called by the probes before / after their timed phases:
a traversal launch that is 10-100x slower than the same launch a minute earlier should show up here as somebody else's
load, somebody else's VRAM, or this process's buffers sitting in GTT (host memory) instead of VRAM."""
import glob, os


def _read(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def snapshot():
    out = []
    for dev in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        vals = {k: _read(os.path.join(dev, k)) for k in ('gpu_busy_percent', 'mem_busy_percent', 'mem_info_vram_used', 'mem_info_vram_total',
                                                          'mem_info_gtt_used', 'mem_info_vis_vram_used')}
        if vals['mem_info_vram_total'] is None:
            continue
        gb = lambda v: -1.0 if v is None else int(v) / 2**30
        out.append('%s: busy %s%% mem-busy %s%% vram %.2f/%.0f GB gtt %.2f GB' % (dev.split('/')[4], vals['gpu_busy_percent'], vals['mem_busy_percent'],
                                                                               gb(vals['mem_info_vram_used']), gb(vals['mem_info_vram_total']),
                                                                               gb(vals['mem_info_gtt_used'])))
    procs = sorted(os.path.basename(p) for p in glob.glob('/sys/class/kfd/kfd/proc/[0-9]*'))
    mine = str(os.getpid())
    own = []
    for f in sorted(glob.glob('/sys/class/kfd/kfd/proc/%s/vram_*' % mine)):
        v = _read(f)
        if v and v != '0':
            own.append('%s=%.2f GB' % (os.path.basename(f), int(v) / 2**30))
    return ' | '.join(out) + ' | kfd processes: %d (%s)%s' % (len(procs), ' '.join(procs[:12]), ' | own ' + ' '.join(own) if own else '')


def cpu_state():
    """CPU side of the box: cgroup quota / throttling counters (v1 and v2 layouts), cpus this process may run on, load average."""
    out = {}
    for path, key in (('/sys/fs/cgroup/cpu.max', 'cpu.max'), ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', 'cfs_quota_us'),
                      ('/sys/fs/cgroup/cpu/cpu.cfs_period_us', 'cfs_period_us')):
        v = _read(path)
        if v is not None:
            out[key] = v
    for path in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        v = _read(path)
        if v:
            for line in v.splitlines():
                k, _, x = line.partition(' ')
                if k in ('nr_throttled', 'throttled_usec', 'throttled_time', 'nr_periods'):
                    out[k] = int(x)
    try:
        out['cpus_allowed'] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    out['loadavg'] = _read('/proc/loadavg')
    return out


if __name__ == '__main__':
    print(snapshot())
    print(cpu_state())
