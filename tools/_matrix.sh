for r in 1 2; do
for L in tools/ab/libm17hip_k3base.so tools/ab/libm17hip_k3ring.so; do
  M17HIP_LIB=$PWD/$L python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --h2d-steps 0 --config2-steps 0 --bursty-steps 0 --parity-channels 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['ms_per_step_single_stream'], d['roofline']['ms_per_step'], d['roofline']['kernel_ms_per_launch']['dcd'])" $L
done; done
