#!/bin/bash
# Round-6 counter passes (each its own rocprofv3 --pmc run with --kernel-trace only, MI355X_MICROARCH.md):
#   1. HBM traffic of the batched extractor's kernels (FETCH_SIZE / WRITE_SIZE, 2048 images)       -> r6_pmc_extractor.json
#   2. VALU issue of k_fast (SQ_INSTS_VALU ..., GRBM_GUI_ACTIVE, 512 images)                        -> r6_pmc_fast.json
#   3. VALU issue of the round-4 kernels the verdict asked about: k_knn2 and k_sbp_assign_cam (the 4-camera rig sequence
#      through the one-call tracker) and k_pose_opt_vio (same run)                                  -> r6_pmc_kernels.json
# Every JSON carries source_sha16 = sha256 of the kernel's source file: bench.py refuses counters of another source.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
sha() { sha256sum $R/vieo_slam_amd/csrc/$1 | cut -c1-16; }
bash $R/tools/pmc_extractor.sh 2048 > $R/gpurun_out/pmc_ext.log 2>&1
python - $R $(sha orb_extractor.hip) <<'PY'
import json, sys
R, sha = sys.argv[1], sys.argv[2]
d = json.load(open(R + "/gpurun_out/pmc_extractor.json"))
d["source_sha16"] = {"vieo_slam_amd/csrc/orb_extractor.hip": sha}
json.dump(d, open(R + "/gpurun_out/r6_pmc_extractor.json", "w"), indent=1)
tot = sum((2 * v["fetch_kb"] + v["write_kb"]) * v["launches"] for v in d["kernels"].values()) * 1024 / 1e9
print("extractor HBM traffic per %d images: %.2f GB" % (d["images_per_launch"], tot))
PY
run() {  # name, counters..., then "--", then the command
  name=$1; shift
  ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rocprofv3 --pmc "${ctr[@]}" --kernel-trace -d $R/gpurun_out/pmc6_$name -o out -- "$@" > $R/gpurun_out/pmc6_$name.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc6_$name -name "*.db" | head -1) > $R/gpurun_out/pmc6_$name.txt 2>&1
}
run fast_b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -- python $R/tools/run_extract.py 512 3
run fast_d GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU -- python $R/tools/run_extract.py 512 3
# round 6: where the cycles that are not vector issue go (verdict item 4): wait / stall and LDS counters of the same launch
run fast_e SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python $R/tools/run_extract.py 512 3
run fast_f SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -- python $R/tools/run_extract.py 512 3
run fast_g SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_IFETCH SQ_ACTIVE_INST_MISC -- python $R/tools/run_extract.py 512 3
run rig_b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -- python $R/tools/run_rig_sequence.py kb8 4 1500 24 5
run rig_d GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python $R/tools/run_rig_sequence.py kb8 4 1500 24 5
python - $R $(sha orb_extractor.hip) $(sha matching.hip) $(sha proj_search.hip) $(sha pose_opt_vio.hip) <<'PY'
import json, re, sys
R, s_orb, s_match, s_sbp, s_pose = sys.argv[1:6]
def read(name):
    out, k = {}, None
    for line in open("%s/gpurun_out/pmc6_%s.txt" % (R, name)):
        if not line.startswith(" "):
            k = line.strip()
            out.setdefault(k, {})
        else:
            m = re.match(r"\s+(\S+)\s+avg (\S+)\s+\(n=(\d+)\)", line)
            if m:
                out[k][m.group(1)] = float(m.group(2))
    return out
def pick(d, sub):
    ks = [k for k in d if sub in k]
    return d[ks[0]] if ks else {}
fb, fd = read("fast_b"), read("fast_d")
c = dict(pick(fb, "k_fast")); c.update(pick(fd, "k_fast"))
for extra in ("fast_e", "fast_f", "fast_g"):
    try:
        c.update(pick(read(extra), "k_fast"))
    except OSError:
        pass
valu = c.get("SQ_INSTS_VALU", 0) / 32.0
cyc = c.get("GRBM_GUI_ACTIVE", 0)
fast = {"source": "tools/pmc_round6.sh: rocprofv3 --pmc in separate passes over a 512-image extraction; averages per (XCD, shader engine) "
                  "counter instance = 8 CUs = 32 SIMDs and per launch",
        "kernel": "k_fast", "images_per_launch": 512, "counters": c, "valu_insts_per_simd": valu, "kernel_cycles": cyc,
        "valu_insts_per_cycle_per_simd": valu / cyc if cyc else None,
        "lane_fill": c.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * c["SQ_ACTIVE_INST_VALU"] * 4 / 4) if c.get("SQ_ACTIVE_INST_VALU") else None,
        "issue_cycles_per_valu_inst": {"full_rate": 1.9, "half_rate": 3.4, "source": "tools/ubench/valu_rate.hip on gfx950 (profiles/r2_valu_rate.txt)"},
        "source_sha16": {"vieo_slam_amd/csrc/orb_extractor.hip": s_orb}}
# the same two passes saw the fused blur + descriptor kernel
cd_ = dict(pick(fb, "k_describe_fused")); cd_.update(pick(fd, "k_describe_fused"))
if cd_:
    fast["k_describe_fused"] = {"counters": cd_, "valu_insts_per_simd": cd_.get("SQ_INSTS_VALU", 0) / 32.0,
                                "kernel_cycles": cd_.get("GRBM_GUI_ACTIVE", 0),
                                "valu_insts_per_cycle_per_simd": (cd_.get("SQ_INSTS_VALU", 0) / 32.0 / cd_["GRBM_GUI_ACTIVE"]) if cd_.get("GRBM_GUI_ACTIVE") else None,
                                "lds_insts_per_simd": cd_.get("SQ_INSTS_LDS", 0) / 32.0}
json.dump(fast, open(R + "/gpurun_out/r6_pmc_fast.json", "w"), indent=1)
rb, rd = read("rig_b"), read("rig_d")
ker = {}
for name, sub, src, sha in (("k_knn2_mfma", "k_knn2_mfma", "matching.hip", s_match), ("k_sbp_assign_cam", "k_sbp_assign_cam", "proj_search.hip", s_sbp),
                            ("k_pose_opt_vio<256, rig>", "k_pose_opt_vio", "pose_opt_vio.hip", s_pose), ("k_fe_fill", "k_fe_fill", "fisheye_stereo.hip", "")):
    c = dict(pick(rb, sub)); c.update(pick(rd, sub))
    if not c:
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0)
    ker[name] = {"counters": c, "source_file": src, "source_sha16": sha,
                 "valu_insts_per_cycle_per_busy_simd_group": (c.get("SQ_INSTS_VALU", 0) / c["SQ_BUSY_CYCLES"]) if c.get("SQ_BUSY_CYCLES") else None,
                 "valu_active_over_busy": (c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_BUSY_CYCLES"]) if c.get("SQ_BUSY_CYCLES") else None,
                 "launch_cycles": cyc}
json.dump({"how": "tools/pmc_round6.sh: the 4-camera KB8 rig sequence (24 frames, one vieo_track_frame call per frame) under rocprofv3 --pmc, "
                  "two passes; averages per counter instance ((XCD, shader engine) = 32 SIMDs) and per launch; SQ_* count quad-cycles",
           "kernels": ker}, open(R + "/gpurun_out/r6_pmc_kernels.json", "w"), indent=1)
print(json.dumps({k: {q: v[q] for q in ("valu_insts_per_cycle_per_busy_simd_group", "valu_active_over_busy", "launch_cycles")} for k, v in ker.items()}, indent=1))
print("k_fast valu/cycle/simd", fast["valu_insts_per_cycle_per_simd"])
PY
# ---- 4. MFMA-busy counters of k_lba_schur (the 205-window batch of a bench step), stamped with lba.hip's hash: bench.py quotes
#         them only while the kernel source is the one they were measured on
bash $R/tools/pmc_lba_schur.sh 205 > $R/gpurun_out/pmc6_lba_schur.log 2>&1
python - $R $(sha lba.hip) <<'PY'
import json, re, sys
R, sha = sys.argv[1], sys.argv[2]
ker, k = {}, None
for line in open(R + "/gpurun_out/pmc_lba_schur_all.txt"):
    if not line.startswith(" "):
        k = line.strip() if "k_lba_schur" in line else None
        if k:
            k = "k_lba_schur<true>" if "<true>" in k else "k_lba_schur<false>"
            ker[k] = {"counters": {}}
    elif k:
        m = re.match(r"\s+(\S+)\s+avg (\S+)\s+\(n=(\d+)\)", line)
        if m:
            ker[k]["counters"][m.group(1)] = float(m.group(2))
what = {"k_lba_schur<false>": "diagonal tiles of the reduced visual system", "k_lba_schur<true>": "off-diagonal tiles (windows with more than 64 rows: the bLarge ones)"}
for k, v in ker.items():
    c = v["counters"]
    if c.get("GRBM_GUI_ACTIVE"):
        v["mfma_busy_fraction_of_simd_time"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32.0 * c["GRBM_GUI_ACTIVE"])
    v["what"] = what[k]
json.dump({"source": "tools/pmc_round6.sh -> tools/pmc_lba_schur.sh 205: rocprofv3 --pmc (own pass, kernel trace only), the r3 bench step's 205 "
                     "visual-inertial windows (40 fixed key frames, every 4th window bLarge) in one lock-step call; averages per counter instance and launch",
           "unit_note": "one SQ counter instance = one (XCD, shader engine) pair = 8 CUs = 32 SIMDs; GRBM_GUI_ACTIVE = the launch's duration in clock "
                        "cycles; MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (32 x GRBM_GUI_ACTIVE)",
           "kernels": ker, "source_sha16": {"vieo_slam_amd/csrc/lba.hip": sha}}, open(R + "/gpurun_out/r6_pmc_lba_schur.json", "w"), indent=1)
print({k: v.get("mfma_busy_fraction_of_simd_time") for k, v in ker.items()})
PY
# ---- 5. k_pose_opt_vio<256> in the C++ replay: instruction fetch, waits, instruction mix
bash $R/tools/pmc_pose.sh r6 > $R/gpurun_out/pmc6_pose.log 2>&1; tail -4 $R/gpurun_out/pmc6_pose.log
