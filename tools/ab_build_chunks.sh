cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export LD_LIBRARY_PATH=$PWD/ab/old; else unset LD_LIBRARY_PATH; fi
  ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v ms_per_frame', r['ms_per_frame'], 'track call', r['ms_track_call'], 'lba', r['ms_per_local_ba'], r['lba_windows']['mean_observations'])"
done; done
