mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
L=tools/ab/libm17hip_k1asm.so
for ph in 0 1 2 3; do
  echo "== placeholders $ph, round-4 K1 (key 11 = 0)"; M17_BISECT_PLACEHOLDERS=$ph M17_BISECT_TUNE=11=0 python tools/bisect_bench.py --rounds 1 --single $L
  for g in 256 0; do
    echo "== placeholders $ph, skew asm, grid $g"; M17_BISECT_PLACEHOLDERS=$ph M17_BISECT_TUNE=13=$g python tools/bisect_bench.py --rounds 1 --single $L
  done
done
