#!/bin/bash
# HBM traffic of the batched extractor's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md 'HBM': kernel trace only next to --pmc; FETCH_SIZE counts 128-B requests as 64 B on gfx950,
# so HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE, both in KB).  Writes gpurun_out/pmc_extractor.json in the layout
# bench.py's pmc_traffic() reads.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-2048}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_ext_$c -o out -- python $R/tools/run_extract.py $B 3 > $R/gpurun_out/pmc_ext_$c.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_ext_$c -name "*.db" | head -1) > $R/gpurun_out/pmc_ext_$c.txt 2>&1
done
python - $R $B <<'PY'
import json, re, sys
R, B = sys.argv[1], int(sys.argv[2])
def read(c):
    out, k = {}, None
    for line in open("%s/gpurun_out/pmc_ext_%s.txt" % (R, c)):
        if not line.startswith(" "):
            k = line.strip()
        else:
            m = re.match(r"\s+%s\s+avg (\S+)\s+\(n=(\d+)\)" % c, line)
            if m:
                out[k] = (float(m.group(1)), int(m.group(2)))
    return out
F, W = read("FETCH_SIZE"), read("WRITE_SIZE")
names = {"fast": ["k_fast"], "blur": ["k_blur"], "pyramid": ["k_resize2", "k_resize"], "describe": ["k_describe_fused", "k_describe"], "quadtree": ["k_quadtree"]}
CALLS = 3  # run_extract.py B 3: three extractions per run
# one counter row per launch (rocprofv3 sums the instances): a kernel's average x its launches per extraction, summed over the
# kernels of a stage (the pyramid is k_resize2 x 3 + k_resize x 1 since round 6) = KB per extraction
def match(k, subs):  # (rocpd_pmc.py prints "vieo::k_name": the base name decides, exactly)
    return k.split("::")[-1].split("<")[0].strip() in subs
kern = {}
for short, subs in names.items():
    f = sum(v[0] * v[1] / CALLS for k, v in F.items() if match(k, subs))
    w = sum(v[0] * v[1] / CALLS for k, v in W.items() if match(k, subs))
    n = sum(v[1] / CALLS for k, v in F.items() if match(k, subs))
    if f or w:
        kern[short] = {"fetch_kb": f, "write_kb": w, "launches": 1, "launches_per_extraction": n}
out = {"how": "tools/pmc_extractor.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate "
              "passes) -- python tools/run_extract.py %d 3; per-kernel averages over the launches; KB per "
              "launch of %d images 752x480; HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024" % (B, B),
       "images_per_launch": B, "kernels": kern,
       "raw_rows": {"FETCH_SIZE": {k: v for k, v in F.items()}, "WRITE_SIZE": {k: v for k, v in W.items()}}}
json.dump(out, open("%s/gpurun_out/pmc_extractor.json" % R, "w"), indent=1)
tot = sum((2 * v["fetch_kb"] + v["write_kb"]) * v["launches"] for v in kern.values()) * 1024 / 1e9
print(json.dumps(kern, indent=1)); print("total GB per %d images: %.2f" % (B, tot))
PY
