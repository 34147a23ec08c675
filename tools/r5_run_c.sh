#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_replay_modes.py -x -q -m gpu -s > gpurun_out/r5c_modes.log 2>&1; echo "rc=$?" >> gpurun_out/r5c_modes.log
python -m pytest tests/test_replay.py -x -q -m gpu -s -k steady_state > gpurun_out/r5c_steady.log 2>&1; echo "rc=$?" >> gpurun_out/r5c_steady.log
python -m pytest tests/test_resident_frame.py tests/test_dropin_replay.py -x -q -m gpu > gpurun_out/r5c_dropin.log 2>&1; echo "rc=$?" >> gpurun_out/r5c_dropin.log
( time python bench.py > gpurun_out/bench_r5c.log 2> gpurun_out/bench_r5c.err ) 2> gpurun_out/bench_r5c.time
tail -1 gpurun_out/bench_r5c.log > gpurun_out/r5c_bench_line.json
tail -4 gpurun_out/r5c_modes.log gpurun_out/r5c_steady.log gpurun_out/r5c_dropin.log; cat gpurun_out/bench_r5c.time; tail -5 gpurun_out/bench_r5c.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5c_bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
s=d["single_stream"]; print({k:s[k] for k in ("frames","ms_per_frame","ms_per_frame_last_200","ms_per_frame_all_runs","ms_per_frame_tracking_call","ate_vs_oracle_m","lba_windows","vs_cpu_single_stream")})
print("drop_in", {k:s["drop_in"][k] for k in ("ms_per_frame","ms_per_frame_last_200","ms_per_frame_host_pointer_form","ate_vs_oracle_m","stage_ms_per_frame")})
print("conc", s["concurrent_trackers"])
for k,v in d.get("single_stream_rig",{}).items():
    if isinstance(v,dict) and "frames" in v: print(k, {q:v[q] for q in ("frames","ms_per_frame_tracking_call","ms_per_frame_tracking_call_gpu","ms_per_frame_python_loop","ate_vs_oracle_m","map_points","lba_windows")})
    else: print(k, str(v)[:600])
v=d.get("single_stream_vision_only"); print("vision", v if not isinstance(v,dict) or "frames" not in v else {q:v[q] for q in ("frames","ms_per_frame_tracking_call","ate_vs_oracle_m","map_points")})
print("inclusive", d.get("inclusive")); print("pcie", d.get("pcie_inclusive"))
PY
