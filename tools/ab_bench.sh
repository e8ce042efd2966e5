#!/bin/bash
# Same-box A/B of two library builds: bench.py's regimes (quick form) alternating A B A B, plus the deferred decode's time alone.
# Usage: ab_bench.sh <libA.so> <libB.so> [rounds]
cd "$(dirname "$0")/.." || exit 1
A=$1; B=$2; R=${3:-2}
out=gpurun_out/r6/ab_$(basename $A .so)_vs_$(basename $B .so).txt; mkdir -p gpurun_out/r6; : > $out
for r in $(seq $R); do
  for lib in $A $B; do
    M17HIP_LIB=$PWD/$lib python3 bench.py --bursty-steps 0 --config2-steps 0 --h2d-steps 0 --cpu-seconds 0 --parity-channels 16 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib', 'value', j['value'], 'ms', j['ms_per_step'], 'single', j['value_single_stream'], j['ms_per_step_single_stream'], 'one-at-a-time', j['roofline']['ms_per_step'], 'parity', j['config']['parity_vs_oracle_first_channels'], j['single_stream']['parity_vs_oracle_3_runs_first_channels'])" >> $out
    M17HIP_LIB=$PWD/$lib python3 - >> $out <<PY
import sys, os
sys.path.insert(0, "m17-cxx-demod_amd"); sys.path.insert(0, "tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch, m17hip, oracle_lib as ol
C, T = 4096, 480000
p = ol.gen_params(seed=20260101, kind=-1, n_frames=T // 1920 - 6, lead_in=3072, noise_sigma=600.0, tail_sigma=600.0, lead_sigma=40000.0, total=T)
c = m17hip.Context(C, T); c.synth(p, C, T)
for i in range(3):
    c.reset(); c.run(); c.frames_count()
c.timing(True); c.timing_reset()
for i in range(4):
    c.reset(); c.run(); c.frames_count()
print("   $lib decode ms per run, one run at a time:", round(c.timing_get("decode")[0] / 4, 3), "demod_seq", round(c.timing_get("demod_seq")[0] / 4, 3))
PY
  done
done
cat $out
