python bench.py > gpurun_out/bench_r4q.log 2>gpurun_out/bench_r4q.err; tail -1 gpurun_out/bench_r4q.log > gpurun_out/r4q_bench_line.json
python - <<PY
import json
d=json.load(open("gpurun_out/r4q_bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
s=d["single_stream"]; print(s["ms_per_frame"], s["ms_per_frame_all_runs"], s["ms_per_frame_tracking_call"], s["ms_per_frame_tracking_call_gpu"], s["vs_cpu_threaded"], s["ms_per_frame_local_ba_inline"], s["vs_cpu_single_stream"])
for k,v in d["single_stream_rig"].items():
    if isinstance(v,dict): print(k, v.get("ms_per_rig_frame"), v.get("ms_per_rig_frame_gpu"), v.get("ms_per_case"))
print(d["single_stream_vision_only"]["ms_per_frame"], d["rig_batch"].get("rig_frames_per_s"), d["rig_batch"].get("ms_per_step"))
print(d["stage_ms_per_step_stream0"]); print(d["parity_sample"]["matches_equal"], d["parity_sample"]["max_se3_error"])
PY
bash tools/prof_rig_tracker.sh r4q | tail -1
