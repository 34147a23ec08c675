cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2; do
for cps in 8 4 2 16; do
  VIEO_LBA_CPS=$cps ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('cps $cps ms_per_frame', r['ms_per_frame'], 'lba', r['ms_per_local_ba'])"
done; done
