#!/usr/bin/env python3
"""profiles/valu.json from the `clk` passes of tools/profile_round.sh: VALU instructions per step of every kernel of the chain (SQ_INSTS_VALU is
reported per hardware instance: average x instances = per dispatch), each kernel's duration ALONE and the clock the chip holds under it
(GRBM_GUI_ACTIVE / duration), K1 alone in both forms, and the shader clock in the real two-batch mix (tools/clock_probe.hip).
bench.py's `roofline.valu` reads it.
Usage: make_valu.py <dir with clk_summary.md, clk_k1_summary.md, clock_probe_default.tsv> <channels> <samples> <runs in the clk pass> <out.json>"""
import json, re, sys
d, C, T, runs, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]

def table(path):
    rows, pmc = {}, {}
    for l in open(path):
        if l.startswith('| _ZN') or l.startswith('| __amd'):
            f = [x.strip() for x in l.strip().strip('|').split('|')]
            rows[f[0]] = {'calls': int(f[1]), 'total_ms': float(f[2]), 'avg_ms': float(f[3])}
        elif l.startswith('- '):
            name, rest = l[2:].split(': ', 1)
            pmc[name] = {m.group(1): (float(m.group(2)), int(m.group(3))) for m in re.finditer(r'(\w+)=([0-9.e+]+)x(\d+)', rest)}
    return rows, pmc

K = {'fir_rrc150': ('fir_rrc150_',), 'dcd': ('dcd_kernel', 'dcd_pipe_kernel'), 'limit_track': ('limit_track_kernel',), 'demod_seq': ('demod_wave_kernel',),
     'decode_deferred': ('decode_deferred_kernel',)}
rows, pmc = table(f'{d}/clk_summary.md')
j = {'channels': C, 'samples': T, 'runs_in_pass': runs, 'kernels': {},
     'source': 'rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES ... --kernel-trace (tools/profile_round.sh, pass clk: one step strictly after the '
               'other; dispatches are serialised under counter collection, so every kernel is measured ALONE); SQ counters are averages over 32 '
               'instances (x32 = per dispatch), GRBM_GUI_ACTIVE over 8 (per XCD: cycles)'}
tot = 0.0
for k, pats in K.items():
    for name in rows:
        if any(p in name for p in pats) and name in pmc:
            r, c = rows[name], pmc[name]
            insts = c['SQ_INSTS_VALU'][0] * c['SQ_INSTS_VALU'][1]
            per_step = insts * r['calls'] / runs
            tot += per_step
            j['kernels'][k] = {'kernel': name.split('(')[0], 'launches_per_step': r['calls'] / runs, 'ms_alone_avg': r['avg_ms'],
                               'valu_insts_per_launch': insts, 'valu_insts_per_step': per_step, 'valu_insts_per_wave': c['SQ_INSTS_VALU'][0] / max(c['SQ_WAVES'][0], 1e-9),
                               'gui_active_cycles_per_launch': c['GRBM_GUI_ACTIVE'][0], 'clock_ghz_alone': c['GRBM_GUI_ACTIVE'][0] / (r['avg_ms'] * 1e6)}
j['valu_insts_per_step'] = tot
try:
    r1, p1 = table(f'{d}/clk_k1_summary.md')
    j['k1_alone_whole_run'] = {}
    for name in r1:
        if 'fir_rrc150' in name and name in p1:
            j['k1_alone_whole_run'][name.split('(')[0]] = {'ms': r1[name]['avg_ms'], 'clock_ghz': p1[name]['GRBM_GUI_ACTIVE'][0] / (r1[name]['avg_ms'] * 1e6),
                                                          'valu_insts': p1[name]['SQ_INSTS_VALU'][0] * p1[name]['SQ_INSTS_VALU'][1]}
except Exception as e:   # noqa: BLE001
    j['k1_alone_whole_run'] = str(e)
try:   # the two-batch regime of the default command with the probe beside it: the busy stretch = windows below 99 % of the idle clock
    lines = open(f'{d}/clock_probe_default.tsv').read().splitlines()
    e0 = float(re.search(r'epoch at launch ([0-9.]+)', lines[0]).group(1))
    pts = [tuple(float(x) for x in l.split()) for l in lines if l and l[0] != '#' and len(l.split()) == 2]
    m = re.search(r'timed region: epoch ([0-9.]+) \.\. ([0-9.]+)', open(f'{d}/clock_probe_default.err').read())
    a, b = float(m.group(1)) - e0, float(m.group(2)) - e0          # the timed region on the probe's time axis (s)
    pts = [(t, c) for t, c in pts if 300.0 < c < 4000.0]            # (a window in which the probe's wave was descheduled or a counter wrapped is not a clock)
    busy = [c for t, c in pts if a + 0.05 <= t / 1e3 <= b - 0.05]
    idle = [c for t, c in pts if t / 1e3 < a - 3.0]
    j['clock_in_mix'] = {'idle_mhz': sum(idle) / len(idle) if idle else None, 'busy_windows': len(busy), 'busy_mean_mhz': sum(busy) / len(busy) if busy else None,
                         'busy_min_mhz': min(busy) if busy else None, 'busy_max_mhz': max(busy) if busy else None,
                         'source': 'tools/clock_probe.hip (one wave of another process: s_memtime against the 100 MHz wall clock, 10 ms windows) beside the timed region '
                                   'of `bench.py --steps 200` (two batches in flight)'}
except Exception as e:   # noqa: BLE001
    j['clock_in_mix'] = str(e)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # (kernel_source_sha16: the stamp bench.py compares with the sources of the build it runs)
j['kernel_source_sha16'] = bench.kernel_source_sha16()
json.dump(j, open(out, 'w'), indent=1)
print(json.dumps({k: (round(v['valu_insts_per_step'] / 1e9, 3), round(v['clock_ghz_alone'], 3)) for k, v in j['kernels'].items()}), 'total G', round(tot / 1e9, 3), j.get('clock_in_mix'))
