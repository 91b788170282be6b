#!/usr/bin/env python3
"""Checks integration/mptrac_hip_glue.c against the reference's src/mptrac.h: every member of ctl_t, met_t,
clim_t, cache_t and atm_t that the glue names must exist there under that name (and every member of
mphip_ctl_t must be filled).  Needs /root/reference (this container only).

  python integration/check_glue_fields.py [/root/reference]
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def struct_members(header, name):
    """member names of `typedef struct { ... } name;` (nested braces not expected in these structs)"""
    end = re.search(r"\}\s*%s\s*;" % name, header)
    if not end:
        raise SystemExit("struct %s not found" % name)
    start = header.rfind("typedef struct", 0, end.start())
    body = header[header.index("{", start) + 1:end.start()]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    names = set()
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("#"):
            continue
        # drop the type: last identifiers before optional [..] groups, comma separated
        decl = re.sub(r"^\s*(const\s+)?(unsigned\s+)?[A-Za-z_]\w*\s+", "", decl, count=1)
        for part in decl.split(","):
            mm = re.match(r"\s*\*?\s*([A-Za-z_]\w*)", part)
            if mm:
                names.add(mm.group(1))
    return names


def macro_list(src, macro):
    m = re.search(r"#define %s\(X\)(.*?)\n\n" % macro, src, re.S)
    if not m:
        m = re.search(r"#define %s\(X\)((?:.*\\\n)*.*\n)" % macro, src)
    return re.findall(r"X\(([^)]*)\)", m.group(1))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    header = open(os.path.join(ref, "src", "mptrac.h"), errors="ignore").read()
    glue = open(os.path.join(HERE, "mptrac_hip_glue.c")).read()
    abi = open(os.path.join(os.path.dirname(HERE), "include", "mptrac_hip.h")).read()
    ctl, met = struct_members(header, "ctl_t"), struct_members(header, "met_t")
    clim, cache, atm = struct_members(header, "clim_t"), struct_members(header, "cache_t"), struct_members(header, "atm_t")
    missing = []
    same = [x.strip() for x in macro_list(glue, "HIP_CTL_SAME_NAME")]
    missing += ["ctl_t." + f for f in same if f not in ctl]
    mq = [x.split(",")[1].strip() for x in macro_list(glue, "HIP_CTL_METEO_QNT")]
    missing += ["ctl_t.qnt_" + f for f in mq if "qnt_" + f not in ctl]
    for f in re.findall(r"\bc->(\w+)", glue):
        if f not in ctl and not f.startswith("f") and f != "qnt_":
            missing.append("ctl_t." + f)
    for macro in ("HIP_MET_3D", "HIP_MET_2D"):
        for x in macro_list(glue, macro):
            f = x.split(",")[1].strip()
            if f not in met:
                missing.append("met_t." + f)
    for f in re.findall(r"\bm->(\w+)", glue):
        if f not in met and f != "f":
            missing.append("met_t." + f)
    for f in re.findall(r"\bclim->(\w+)", glue):
        if f not in clim:
            missing.append("clim_t." + f)
    for f in re.findall(r"\bcache->(\w+)", glue):
        if f not in cache:
            missing.append("cache_t." + f)
    for f in re.findall(r"\batm->(\w+)", glue):
        if f not in atm:
            missing.append("atm_t." + f)
    # the other direction: every member of mphip_ctl_t is filled by hip_ctl
    dev = struct_members(abi, "mphip_ctl_t")
    filled = set(same) | {"qnt_met", "wet_depo_pre", "wet_depo_ic_h", "wet_depo_bc_h"}
    if len(re.findall(r"d->qnt_tracer\[MPHIP_TR_\w+\] = c->qnt_C\w+;", glue)) == 5:
        filled.add("qnt_tracer")
    unfilled = sorted(f for f in dev if f not in filled and not f.startswith("pad"))
    # the module_meteo quantities in ABI order
    order = [e.lower() for e in re.findall(r"MPHIP_MQ_(\w+)", abi.split("MPHIP_NMQ")[0])]
    glue_order = [x.split(",")[0].strip().lower() for x in macro_list(glue, "HIP_CTL_METEO_QNT")]
    problems = sorted(set(missing))
    if problems:
        print("members the glue names that the reference does not have:", ", ".join(problems))
    if unfilled:
        print("members of mphip_ctl_t the glue leaves unset:", ", ".join(unfilled))
    if sorted(order) != sorted(glue_order):
        print("module_meteo quantity list differs from the MPHIP_MQ_* enumerators")
    ok = not problems and not unfilled and sorted(order) == sorted(glue_order)
    print("checked %d ctl_t members, %d meteo quantities, %d met_t fields: %s" % (
        len(same), len(mq), len(macro_list(glue, "HIP_MET_3D")) + len(macro_list(glue, "HIP_MET_2D")), "ok" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
