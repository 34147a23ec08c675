cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2; do
VIEO_LBA_TIMING=1 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 2>/tmp/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('ms_per_frame', r['ms_per_frame'], 'lba', r['ms_per_local_ba'], 'job', r['ms_per_local_mapping_job'], r['caller_ms_per_frame'])"
python - <<PY
import re
st=[];ro=[];wa=[]
for l in open('/tmp/err.txt'):
    m=re.search(r'staging ([\d.]+) ms, (\d+) rounds ([\d.]+) ms \(of which waiting for the stream ([\d.]+)\)',l)
    if m: st.append(float(m.group(1))); ro.append(float(m.group(3))); wa.append(float(m.group(4)))
import statistics as S
print('lba_run: staging %.3f rounds %.3f waiting %.3f (mean of %d)'%(S.mean(st),S.mean(ro),S.mean(wa),len(st)))
PY
done
