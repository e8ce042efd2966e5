#!/bin/bash
# Round profile on the GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) and, in SEPARATE passes,
# the HBM traffic counters (FETCH_SIZE / WRITE_SIZE need 3 + 2 of the 4 TCC slots) and SQ issue/stall counters.
# Usage (from the repo root, inside gpurun):  bash tools/profile_round.sh r1
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the profiler's preloaded library initialises the HIP runtime before python runs: the queue count must be in the environment already)
export GPU_MAX_HW_QUEUES=16
# ONE step strictly after the other (--in-flight 1, no single-stream leg, no extra one-at-a-time pass): every launch in the trace is in
# the regime bench.py's `roofline` object is computed from, so the kernel_trace_stats average and the line's kernel_ms_per_launch agree
ARGS="--in-flight 1 --single-stream 0 --one-at-a-time 0 --steps 4 --warmup 1 --prewarm 0 --cpu-seconds 0 --parity-channels 0 --h2d-steps 0 --config2-steps 0 --bursty-steps 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
# (the instruction counts go into profiles/valu.json, which prices the HEADLINE regime — two batches in flight, ten equal segments per run: the ramp of short
#  segments the library gives a run that is alone in flight, m17hip_tune key 33 = -1, is pinned off for the two counter passes that feed it)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $OUT/sq -o sq -- python3 $R/bench.py $ARGS --tune 33=0 > $OUT/sq.log 2>&1
# the clock the chip holds: GRBM_GUI_ACTIVE (cycles the GPU was active during a dispatch) / the dispatch's duration, with the VALU-busy counter
# beside it — per kernel of the chain (dispatches are serialised under counter collection: each kernel ALONE), and for K1 alone in both forms
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $OUT/clk -o clk -- python3 $R/bench.py $ARGS --tune 33=0 > $OUT/clk.log 2>&1
M17HIP_LIB=$R/m17-cxx-demod_amd/libm17hip_tools.so rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $OUT/clk_k1 -o clk_k1 -- python3 $R/tools/k1_only.py > $OUT/clk_k1.log 2>&1
# ... and in the real mix (no profiler): one wave of another process records shader cycles against the 100 MHz wall clock (tools/clock_probe.hip)
# while the default command's two-batch regime runs (200 timed steps = 4 s)
$R/tools/clock_probe 40 > $OUT/clock_probe_default.tsv 2>&1 &
python3 $R/bench.py --steps 200 --warmup 5 --prewarm 100 --cpu-seconds 0 --parity-channels 0 --h2d-steps 0 --config2-steps 0 --bursty-steps 0 --single-stream 0 --one-at-a-time 0 > $OUT/clock_probe_default.log 2> $OUT/clock_probe_default.err
wait
# the default command (two batches in flight, then the single-stream regime), for the record
rocprofv3 --kernel-trace --stats -d $OUT/trace_default -o trace_default -- python3 $R/bench.py --steps 4 --warmup 1 --prewarm 4 --cpu-seconds 0 --parity-channels 0 --h2d-steps 0 --config2-steps 0 --bursty-steps 0 > $OUT/trace_default.log 2>&1
# BASELINE configs[1] (the `config2` object of the default line): FIR + correlator, 1024 x 480 000
rocprofv3 --kernel-trace --stats -d $OUT/trace_config2 -o trace_config2 -- python3 $R/bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 0 --parity-channels 0 > $OUT/trace_config2.log 2>&1
for p in trace fetch write sq clk clk_k1 trace_default trace_config2; do
  python3 $R/tools/rocpd_summary.py $OUT/$p/${p}_results.db $OUT/${p}_summary.md > /dev/null 2>&1
  grep -h '"metric"' $OUT/$p.log | head -1 > $OUT/${p}_bench_line.json
done
ls -la $OUT
