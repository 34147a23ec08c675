#!/usr/bin/env python3
"""After tools/pmc_round6.sh: gpurun_out/r6_pmc_fast.json (k_fast's counters, stamped with orb_extractor.hip's hash) ->
profiles/r6_pmc_fast.json with the derived tables bench.py quotes:
  wave_cycle_accounting   SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY split into shares,
  instruction_classes     tools/fast_issue.py's static class table of the kernel's loops (from the ISA of the same source),
  issue_fraction          measured VALU instructions x the class mix's issue cycles over the kernel's cycles.
usage: python tools/finish_fast_profile.py [gpurun_out/r6_pmc_fast.json] [profiles/r6_pmc_fast.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import fast_issue  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6_pmc_fast.json")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r6_pmc_fast.json")
d = json.load(open(src))
c = d["counters"]
wc = c.get("SQ_WAVE_CYCLES")
if wc:
    act = {"valu": c.get("SQ_ACTIVE_INST_VALU", 0), "scalar": c.get("SQ_ACTIVE_INST_SCA", 0), "lds": c.get("SQ_ACTIVE_INST_LDS", 0),
           "misc": c.get("SQ_ACTIVE_INST_MISC", 0)}
    tot = sum(act.values()) or 1.0
    d["wave_cycle_accounting"] = {
        "note": "SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY (same unit, summed over the resident wavefronts of a "
                "counter instance): where a resident wavefront spends its time",
        "executing_an_instruction": c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "ready_but_waiting_for_the_issue_port": c.get("SQ_WAIT_INST_ANY", 0) / wc,
        "waiting_on_data_or_counters": c.get("SQ_WAIT_ANY", 0) / wc, "of_executing": {k: v / tot for k, v in act.items()},
        "resident_wavefronts_per_simd": wc * 4 / (32.0 * d["kernel_cycles"]) if d.get("kernel_cycles") else None,  # (SQ_* count quad-cycles)
        "lds": {"wait_inst_lds_share_of_wave_cycles": c.get("SQ_WAIT_INST_LDS", 0) / wc,
                "bank_conflict_share_of_lds_active_cycles": (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None}}
json.dump(d, open(dst, "w"), indent=1)  # (fast_issue reads the newest profiles/r*_pmc_fast.json: this one)
res = fast_issue.main()
d["instruction_classes"] = {k: res[k] for k in ("per_pass", "per_cell_instructions_static", "valu_per_cell_static", "valu_half_rate_share",
                                                "issue_cycles_per_valu_class", "measured") if k in res}
for k in ("issue_fraction", "issue_fraction_note", "issue_fraction_with_scalar_sharing_the_port"):
    d[k] = res.get(k)
w = d.get("wave_cycle_accounting", {})
d["verdict"] = ("k_fast is at its roof, and the roof is vector issue: %.2f of a SIMD's cycles are VALU issue (tools/fast_issue.py: measured "
                "instruction count x the loops' class mix), %.2f when the scalar instructions are counted on the same arbiter; a resident wavefront "
                "spends %.0f %% of its time ready but waiting for the port, %.0f %% executing, %.0f %% on data. LDS is not a factor (%.1f %% of wave "
                "cycles). Only fewer instructions can move it."
                % (d["issue_fraction"], d["issue_fraction_with_scalar_sharing_the_port"], 100 * w.get("ready_but_waiting_for_the_issue_port", 0),
                   100 * w.get("executing_an_instruction", 0), 100 * w.get("waiting_on_data_or_counters", 0),
                   100 * w.get("lds", {}).get("wait_inst_lds_share_of_wave_cycles", 0)))
json.dump(d, open(dst, "w"), indent=1)
print("wrote", dst, "issue fraction", d["issue_fraction"])
