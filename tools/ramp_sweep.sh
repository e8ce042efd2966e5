#!/bin/bash
# ramp of segment lengths at the start of a run (m17hip_tune key 33) against the regimes of bench.py (quick form), same box
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r6/ramp_sweep.txt; mkdir -p gpurun_out/r6; : > $out
for r in ${RAMPS:-0 2400 4800 9600 12000 24000 0}; do
  python3 bench.py --bursty-steps 0 --config2-steps 0 --h2d-steps 0 --cpu-seconds 0 --parity-channels 16 --tune 33=$r $EXTRA 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ramp $r', 'value', j['value'], 'ms', j['ms_per_step'], 'single', j['value_single_stream'], j['ms_per_step_single_stream'], 'one-at-a-time', j['roofline']['ms_per_step'], 'parity', j['config']['parity_vs_oracle_first_channels'], j['single_stream']['parity_vs_oracle_3_runs_first_channels'], 'K5 ms/launch', j['roofline']['kernel_ms_per_launch']['demod_seq'])" >> $out
done
cat $out
