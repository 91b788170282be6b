#!/usr/bin/env python3
"""The VALU-issue roof of the step kernel: profiles/alu_model.json from a dynamic instruction mix
(tools/profile_valu_mix.sh -> profiles/<tag>_c3_instruction_mix.txt, counters per 64 particle-steps = per wave-step)
and the issue costs of gfx950 with the SIMDs full (tools/micro/valu_issue.hip -> profiles/r04_valu_issue.txt, the
"4 waves/SIMD" rows, shader cycles per wave-instruction per SIMD).

  python tools/alu_model.py profiles/r04_c3_instruction_mix.txt C3 [sustained clock in GHz, default 2.1]

Classes the counters do not split are priced with the mean of what the disassembly holds for them (stated below);
the sustained clock is GRBM_GUI_ACTIVE / kernel time of the same profile (2.1 GHz under this kernel, not the
nominal 2.4)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = {                      # profiles/r04_valu_issue.txt, 4 waves per SIMD
    "ADD_F64": 4.26, "MUL_F64": 4.31, "FMA_F64": 4.50, "TRANS_F64": 16.3,
    "CVT": 4.30, "INT64": 4.30,
    "INT32": 3.3,             # half plain VOP1/2 adds / logic (2.4), half VOP3 / multiplies (4.2)
    "ADD_F32": 2.8, "MUL_F32": 2.8, "FMA_F32": 2.9, "TRANS_F32": 8.2,
    "OTHER": 3.6,             # moves (2.35), fp64 compares and VOP3 selects (4.2), min / max / ldexp (4.2)
}


def main():
    mix_file, workload = sys.argv[1], sys.argv[2]
    clock = float(sys.argv[3]) if len(sys.argv) > 3 else 2.1
    mix = {}
    for line in open(mix_file):
        f = line.split()
        if len(f) >= 4 and f[0].startswith("SQ_"):
            mix[f[0]] = float(f[-1])
    total = mix["SQ_INSTS_VALU"]
    cycles, named = 0.0, 0.0
    rows = {}
    for cls, cost in COST.items():
        if cls == "OTHER":
            continue
        n = mix.get("SQ_INSTS_VALU_" + cls, 0.0)
        named += n
        cycles += n * cost
        rows[cls] = n
    rows["OTHER"] = total - named
    cycles += rows["OTHER"] * COST["OTHER"]
    out_file = os.path.join(ROOT, "profiles", "alu_model.json")
    out = json.load(open(out_file)) if os.path.exists(out_file) else {}
    out[workload] = {"cycles_per_wave_step": round(cycles, 1), "valu_insts_per_64_particle_steps": round(total, 1),
                     "sustained_clock_ghz": clock, "instructions_by_class": {k: round(v, 1) for k, v in rows.items()},
                     "cycles_per_instruction_by_class": COST,
                     "source": f"profiles/r04_valu_issue.txt x {os.path.relpath(mix_file, ROOT)} (tools/alu_model.py)"}
    json.dump(out, open(out_file, "w"), indent=1)
    print(json.dumps(out[workload], indent=1))


if __name__ == "__main__":
    main()
