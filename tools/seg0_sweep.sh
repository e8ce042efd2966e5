#!/bin/bash
# first-segment length (m17hip_tune key 4, tools build) against the regimes of bench.py (quick form)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r6/seg0_sweep.txt; mkdir -p gpurun_out/r6; : > $out
for s0 in 0 2400 4800 9600 19200; do
  M17HIP_LIB=$PWD/m17-cxx-demod_amd/libm17hip_tools.so python3 bench.py --bursty-steps 0 --config2-steps 0 --h2d-steps 0 --cpu-seconds 0 --parity-channels 16 --tune 4=$s0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('seg0 $s0', 'value', j['value'], 'ms', j['ms_per_step'], 'single', j['value_single_stream'], j['ms_per_step_single_stream'], 'one-at-a-time', j['roofline']['ms_per_step'], 'parity', j['config']['parity_vs_oracle_first_channels'], j['single_stream']['parity_vs_oracle_3_runs_first_channels'])" >> $out
done
cat $out
