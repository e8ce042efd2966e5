#!/bin/bash
# rocprofv3 kernel traces of the continued-stream regime alone, in both call orders: the last ~2 steps with queue ids
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/r6
for order in new old; do
  rm -rf /tmp/st_$order
  ORDER=$order rocprofv3 --kernel-trace -d /tmp/st_$order -o t -- python3 $R/tools/stream_only.py 0 > $R/gpurun_out/r6/stream_only_$order.log 2>&1
  db=$(find /tmp/st_$order -name '*_results.db' | head -1)
  python3 $R/tools/rocpd_queues.py $db ${NDISP:-260} > $R/gpurun_out/r6/stream_timeline_$order.txt
  tail -2 $R/gpurun_out/r6/stream_only_$order.log
done
