#!/bin/bash
# N independent sequences on ONE GPU as N PROCESSES (examples/replay_main each): does the device serve several live rigs when
# every tracker has its own process (its own hardware queues), where N trackers in one process do not scale
# (bench: concurrent_trackers)?
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
python tools/write_sequence.py /tmp/seq.vseq --frames 200 > /dev/null
for n in 1 2 4 8 16; do
  t0=$(date +%s.%N)
  for i in $(seq 1 $n); do ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 > /tmp/cp_$i.json & done
  wait
  t1=$(date +%s.%N)
  python - $n $t0 $t1 <<'PY'
import json, sys
n, t0, t1 = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
ms = [json.loads(open("/tmp/cp_%d.json" % i).read().strip().splitlines()[-1])["ms_per_frame"] for i in range(1, n + 1)]
fps = sum(1e3 / m for m in ms)
print("%2d processes: ms per frame of each (timed loop only) mean %.3f max %.3f -> %.0f frames/s in total (wall incl. start-up %.1f s)" % (n, sum(ms) / n, max(ms), fps, t1 - t0))
PY
done
