for r in 1 2; do for g in 0 1280 1024 2048 4096 12288; do
python bench.py --config 2 --steps 5 --warmup 2 --cpu-seconds 0 --parity-channels 0 --tune 13=$g 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 grid', sys.argv[1], d['value'], d['ms_per_step'])" $g
done; done
