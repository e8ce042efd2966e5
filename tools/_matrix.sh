L=m17-cxx-demod_amd/libm17hip.so
for q in 8 16 24 32 48; do
for ph in 0 3; do
  echo "== GPU_MAX_HW_QUEUES $q placeholders $ph"; GPU_MAX_HW_QUEUES=$q M17_BISECT_PLACEHOLDERS=$ph python tools/bisect_bench.py --rounds 1 --single $L
done; done
