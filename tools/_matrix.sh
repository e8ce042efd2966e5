for i in 1 2; do
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --h2d-steps 0 --config2-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench default', d['value'], d['ms_per_step'], d['value_single_stream'], d['ms_per_step_single_stream'], d['roofline']['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --h2d-steps 0 --config2-steps 0 --tune 20=1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20=1', d['value'], d['ms_per_step'], d['value_single_stream'], d['ms_per_step_single_stream'], d['roofline']['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --h2d-steps 0 --config2-steps 0 --tune 20=0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20=0', d['value'], d['ms_per_step'], d['value_single_stream'], d['ms_per_step_single_stream'], d['roofline']['ms_per_step'])"
python tools/bisect_bench.py --rounds 1 --single m17-cxx-demod_amd/libm17hip.so
M17_BISECT_TUNE=20=1 python tools/bisect_bench.py --rounds 1 m17-cxx-demod_amd/libm17hip.so
done
