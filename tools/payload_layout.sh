#!/bin/bash
# bench.py's regimes with the two call orders of a continued stream (quick form: no cpu / config2 / bursty / h2d legs).  LIB=... picks a library build.
cd "$(dirname "$0")/.." || exit 1
out=${OUT:-gpurun_out/r6/stream_order.txt}
mkdir -p "$(dirname $out)"; : > $out
for rep in 1 2; do
  for order in run_then_fetch fetch_then_run; do
    M17HIP_LIB=${LIB:-$PWD/m17-cxx-demod_amd/libm17hip.so} python3 bench.py --bursty-steps 0 --config2-steps 0 --h2d-steps 0 --cpu-seconds 0 --parity-channels 16 --stream-order $order $EXTRA 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$order', 'value', j['value'], 'ms', j['ms_per_step'], 'single', j['value_single_stream'], j['ms_per_step_single_stream'], 'one-at-a-time', j['roofline']['ms_per_step'], 'parity', j['config']['parity_vs_oracle_first_channels'], j['single_stream']['parity_vs_oracle_3_runs_first_channels'], j['single_stream']['kernel_ms'])" >> $out
  done
done
cat $out
