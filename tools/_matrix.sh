export GPU_MAX_HW_QUEUES=24
for ph in 0 1 2 3; do for pre in 0 1 2 3; do python3 tools/stream_only.py $ph $pre 2>&1 | tail -1; done; done
