B="python bench.py --steps 10 --warmup 4 --cpu-seconds 0 --h2d-steps 0 --config2-steps 0 --parity-channels 0 --bursty-steps 10 --single-stream 0 --one-at-a-time 0"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["bursty"]["ms_per_step"], d["bursty"]["ratio_to_always_on"], d["bursty"]["kernel_ms"])'
$B 2>/dev/null | python3 -c "$P" "auto"
$B --tune 10=1 2>/dev/null | python3 -c "$P" "k3_latency_form"
$B --in-flight 3 2>/dev/null | python3 -c "$P" "inflight3"
$B --in-flight 4 2>/dev/null | python3 -c "$P" "inflight4"
$B --in-flight 4 --tune 10=1 2>/dev/null | python3 -c "$P" "inflight4_k3lat"
$B --tune 3=24000 2>/dev/null | python3 -c "$P" "seg24000"
$B --tune 3=32000 2>/dev/null | python3 -c "$P" "seg32000"
