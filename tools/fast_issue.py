#!/usr/bin/env python3
"""Static instruction-class table of k_fast's loops from its ISA (hipcc -save-temps), weighted by measured trip counts.

    python tools/fast_issue.py [trip counts json]     (writes nothing; prints the table and a JSON block)

The kernel's time follows its vector instruction stream (profiles/r5_pmc_fast.json: traffic 1.13 x algorithmic, 0.278
VALU instructions per cycle and SIMD); round 5 bracketed the issue fraction between "all full rate" (0.46) and "all half
rate" (0.82).  This tool closes the bracket: every VALU / SALU / LDS / VMEM instruction of the kernel's loops is put in
its issue class (tools/ubench/valu_rate.hip on gfx950, profiles/r5_valu_rate.txt: full-rate vector ops 1.7-2.0 cycles
per wavefront instruction and SIMD, half-rate ones 3.3-3.4), the loops are recognised by their content (pass A: the
v_bitop3 compass test, four copies for the four byte alignments; pass B: the min3 / max3 strength network; pass C: the
3 x 3 neighbourhood of byte reads), and a cell's instruction count is the loops' bodies times the trip counts
-DVIEO_FAST_STATS measured on the bench's frames (profiles/r4o_fast_cells.txt).
"""
import json
import re
import subprocess
import sys
import os
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = ("v_perm", "v_pk_", "v_min_", "v_max_", "v_min3", "v_max3", "v_med3", "v_alignbyte", "v_alignbit", "v_bfe", "v_bfi", "v_bcnt",
        "v_mad_", "v_mul_lo", "v_mul_hi", "v_lshlrev", "v_lshrrev", "v_lshl_", "v_add3", "v_or3", "v_and_or", "v_lshl_add", "v_lshl_or", "v_add_lshl",
        "v_cmp", "v_cndmask", "v_sad", "v_dot", "v_mbcnt", "v_readlane", "v_readfirstlane", "v_writelane", "v_xad", "v_cvt")
FULL_EXC = ("v_min_u16", "v_max_u16")  # measured full rate


def classify(op):
    if op.startswith("v_"):
        if op.startswith(FULL_EXC):
            return "valu_full"
        if op.endswith("_dpp") or "dpp" in op:
            return "valu_half"
        return "valu_half" if op.startswith(HALF) else "valu_full"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def kernel_isa():
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                               "-Wno-unused-function", "-Wno-unused-result", "--cuda-device-only", "-save-temps", "-c",
                               os.path.join(ROOT, "vieo_slam_amd", "csrc", "orb_extractor.hip"), "-o", os.path.join(tmp, "x.o")], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = open(os.path.join(tmp, "orb_extractor-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    a = next(i for i, l in enumerate(s) if "Begin function _ZN4vieo6k_fastE" in l)
    b = next(i for i in range(a, len(s)) if "End function" in s[i])
    return s[a:b]


def main():
    lines = kernel_isa()
    # instructions with their position; labels
    inst, labels = [], {}
    for l in lines:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(inst)
            continue
        if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
            continue
        op = t.split()[0]
        dpp = " row_" in t or " quad_perm" in t or "row_bcast" in t
        inst.append((op + ("_dpp" if dpp else ""), t))
    # loops = backward branches
    loops = []
    for i, (op, t) in enumerate(inst):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    # innermost-first attribution: a loop's own body = its range minus the ranges of loops nested in it
    loops = sorted(set(loops), key=lambda ab: ab[1] - ab[0])
    owner = [None] * len(inst)
    for k, (a, b) in enumerate(loops):
        for i in range(a, b + 1):
            if owner[i] is None:
                owner[i] = k
    def table(idx):
        c = {}
        for i in idx:
            c[classify(inst[i][0])] = c.get(classify(inst[i][0]), 0) + 1
        return c
    def tag(idx):
        ops = [inst[i][0] for i in idx]
        n3 = sum(o.startswith(("v_min3", "v_max3")) for o in ops)
        nb = sum(o.startswith("v_bitop3") for o in ops)
        nu8 = sum(o.startswith("ds_read_u8") for o in ops)
        if n3 >= 40:
            return "pass B (strength of 64 survivors)"
        if nb >= 6:
            return "pass A (compass test of 256 pixels)"
        if nu8 >= 8:
            return "pass C (3 x 3 suppression of 64 corners)"
        return None
    out = []
    for k, (a, b) in enumerate(loops):
        idx = [i for i in range(a, b + 1) if owner[i] == k]
        out.append({"loop": k, "first": a, "last": b, "own_instructions": len(idx), "kind": tag(idx), "classes": table(idx)})
    rest = [i for i in range(len(inst)) if owner[i] is None]
    straight = {"own_instructions": len(rest), "classes": table(rest)}
    # ---- weights: per cell trip counts on the bench's frames (profiles/r4o_fast_cells.txt: -DVIEO_FAST_STATS; EuRoC level mix,
    # 700 cells per image, a cell's interior ~1 690 pixels = 6.6 steps of 256 pixels in pass A).  The four alignment copies
    # of pass A and the six inlined copies of pass B have the same mix: one representative (the mean) per pass.
    trips = {"pass A (compass test of 256 pixels)": 6.6, "pass B (strength of 64 survivors)": 4.72, "pass C (3 x 3 suppression of 64 corners)": 1.51}
    if len(sys.argv) > 1:
        trips.update(json.load(open(sys.argv[1])))
    cyc = {"valu_full": 1.9, "valu_half": 3.4}
    per_cell, per_pass = {}, {}
    for kind, t in trips.items():
        copies = [o for o in out if o["kind"] == kind]
        mix = {}
        for o in copies:
            for c, n in o["classes"].items():
                mix[c] = mix.get(c, 0.0) + n / len(copies)
        per_pass[kind] = {"trips_per_cell": t, "copies_in_the_code": len(copies), "instructions_per_trip": {k: round(v, 1) for k, v in mix.items()}}
        for c, n in mix.items():
            per_cell[c] = per_cell.get(c, 0.0) + n * t
    # once per cell: the tile's load loops (the two largest untagged loops: dword loads -> LDS with the halo), the score
    # clear, the cell set-up in the outermost loop
    once = sorted([o for o in out if not o["kind"]], key=lambda o: -o["own_instructions"])
    setup = {}
    for o in once[:4]:
        for c, n in o["classes"].items():
            setup[c] = setup.get(c, 0.0) + n * (1.0 if o["own_instructions"] > 200 else 2.0)  # (the tile loops take ~2 trips)
    per_pass["per cell (tile load, set-up)"] = {"instructions": {k: round(v, 1) for k, v in setup.items()}}
    for c, n in setup.items():
        per_cell[c] = per_cell.get(c, 0.0) + n
    valu_static = per_cell.get("valu_full", 0) + per_cell.get("valu_half", 0)
    half_share = per_cell.get("valu_half", 0) / max(valu_static, 1)
    res = {"per_pass": per_pass, "per_cell_instructions_static": {k: round(v, 1) for k, v in per_cell.items()},
           "valu_per_cell_static": round(valu_static, 1), "valu_half_rate_share": round(half_share, 3), "issue_cycles_per_valu_class": cyc}
    # ---- against the counters: the measured VALU instruction count and kernel cycles per cell and SIMD
    import glob
    pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fast.json")), key=lambda f: int(re.match(r"r(\d+)", os.path.basename(f)).group(1)))
    if pm:
        d = json.load(open(pm[-1]))
        simds = 256 * 4
        cells = d["images_per_launch"] * 700.0 / simds  # cells per SIMD and launch
        valu_meas = d["valu_insts_per_simd"] / cells
        cyc_cell = d["kernel_cycles"] / cells
        c = d["counters"]
        salu_meas = c.get("SQ_INSTS_SALU", 0) / 32.0 / cells
        lds_meas = c.get("SQ_INSTS_LDS", 0) / 32.0 / cells
        mean_cyc = (1 - half_share) * cyc["valu_full"] + half_share * cyc["valu_half"]
        res["measured"] = {"counter_file": os.path.relpath(pm[-1], ROOT), "valu_per_cell": round(valu_meas, 1), "salu_per_cell": round(salu_meas, 1),
                           "lds_per_cell": round(lds_meas, 1), "kernel_cycles_per_cell_per_simd": round(cyc_cell, 1),
                           "static_over_measured_valu": round(valu_static / valu_meas, 3)}
        res["issue_fraction"] = round(valu_meas * mean_cyc / cyc_cell, 3)
        res["issue_fraction_note"] = ("VALU issue cycles of a cell (measured instruction count x the mix's mean %.2f cycles per wavefront instruction, "
                                      "%.0f %% half-rate) over the cycles a SIMD spends per cell; the scalar (%.0f per cell) and LDS (%.0f) instructions "
                                      "issue beside it" % (mean_cyc, 100 * half_share, salu_meas, lds_meas))
        res["issue_fraction_with_scalar_sharing_the_port"] = round((valu_meas * mean_cyc + salu_meas * 1.9) / cyc_cell, 3)
    for o in out:
        print("loop %2d  [%5d..%5d]  own %4d  %-42s %s" % (o["loop"], o["first"], o["last"], o["own_instructions"], o["kind"] or "-", o["classes"]))
    print("straight-line:", straight)
    print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    main()
