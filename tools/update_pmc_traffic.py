#!/usr/bin/env python3
"""Turn the traffic.json of a tools/profile.sh run into profiles/pmc_traffic.json, stamped with the hash of
the kernel sources it was measured on (bench.py only quotes it for that build).

Usage: tools/update_pmc_traffic.py gpurun_out/prof_<tag> <workload> <profiles/summary file it belongs to>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, workload, summary = sys.argv[1], sys.argv[2], sys.argv[3]
raw = json.load(open(os.path.join(src, "traffic.json")))
out = {
    workload: raw["traffic_bytes"],
    "_build_id": bench.build_id(),
    "_how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 10 --warmup 2` "
            "(tools/profile.sh), median step_kernel dispatch, KiB -> bytes; FETCH_SIZE x the factor that brings "
            "pack_kernel's coalesced reads to their known byte count (the gfx950 half-count of MI355X_MICROARCH.md); "
            "WRITE_SIZE as is (it equals the kernel's known 44 B of stores per particle)",
    "_raw": raw,
    "_profile": summary,
}
if "valu_busy_frac" in raw:
    out["_valu_busy_frac"] = {
        workload: round(raw["valu_busy_frac"], 3),
        "_how": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) of a steady step_kernel dispatch "
                "(tools/summarize_prof.py); %.0f VALU instructions per 64 particle-steps"
                % raw.get("valu_insts_per_64_particle_steps", float("nan")),
    }
# share of fp64 arithmetic among the VALU instructions, from the instruction-mix passes of the same build
# (tools/profile_valu_mix.sh -> gpurun_out/prof_<tag>mix/valu_mix.txt), if they were run
mix_file = os.path.join(src.rstrip("/") + "mix", "valu_mix.txt")
if os.path.exists(mix_file):
    mix = {}
    for line in open(mix_file):
        f = line.split()
        if len(f) >= 4 and f[0].startswith("SQ_"):
            mix[f[0]] = float(f[-1])      # (per 64 particle-steps: the last column)
    fp64 = sum(mix.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
    if mix.get("SQ_INSTS_VALU"):
        out["_fp64_valu_frac"] = {
            workload: round(fp64 / mix["SQ_INSTS_VALU"], 3),
            "_how": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 / SQ_INSTS_VALU of a steady step_kernel dispatch "
                    "(tools/profile_valu_mix.sh, same build)",
        }
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
