B="python bench.py --gpus 1 --in-flight 1 --single-stream 0 --one-at-a-time 0 --steps 6 --warmup 2 --cpu-seconds 0 --config2-steps 0 --h2d-steps 0 --parity-channels 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"])'
for i in 1 2; do
echo "default   : $($B 2>/dev/null | python -c "$P")"
echo "tune 10=1 : $($B --tune 10=1 2>/dev/null | python -c "$P")"
done
