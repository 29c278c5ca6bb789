#!/usr/bin/env python3
"""Fold the raw rocprofv3 output of tools/profile_r3.sh (under gpurun_out/<dir>) into the small files that are
committed under profiles/rNN/ and that bench.py reads:

  counters.json         per case of tools/pmc_case.py: the solve / product kernel's SQ counters (mean per launch),
                        FETCH_SIZE / WRITE_SIZE (KB per launch; gfx950: FETCH_SIZE counts 128-B requests at 64 B and is
                        doubled, MI355X_MICROARCH.md), contacts per launch, and the derived figures
                          valu_insts_per_launch      = SQ_INSTS_VALU          (wave-instructions)
                          lane_utilisation           = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
                          hbm_bytes_per_launch       = 2 x FETCH + WRITE
  kernel_stats_<leg>.csv   rocprofv3 --kernel-trace --stats summary of `bench.py --legs <leg>`

usage: tools/profile_summary.py gpurun_out/<dir> [profiles/rNN]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

MAIN = ("step_solve_kernel", "step_quad_kernel", "tile_solve_kernel", "quad_solve_kernel", "patch_solve_kernel", "global_solve_kernel",
        "matvec_tile_kernel", "pair_solve_kernel", "duo_solve_kernel", "chol_", "box_dantzig_kernel", "box_murty_kernel")


def short(name):
    m = re.search(r"namespace\)::(\w+)", name)
    return m.group(1) if m else name.split("(")[0].split("<")[0]


def counters(directory):
    acc = {}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return acc


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else None
    out = {}
    cases = sorted({os.path.basename(d).split("_", 1)[1] for d in glob.glob(os.path.join(src, "sq1_*")) if os.path.isdir(d)})
    for case in cases:
        merged = {}
        for kind in ("sq1", "sq2", "fetch", "write"):
            for k, cs in counters(os.path.join(src, "%s_%s" % (kind, case))).items():
                for c, v in cs.items():
                    # the first launch of a case is the warm-up (cold instruction cache, plan upload): drop it when there are more
                    vals = v[1:] if len(v) > 1 else v
                    merged.setdefault(k, {})[c] = sum(vals) / len(vals)
                    merged[k]["launches_" + kind] = len(v)
        log = os.path.join(src, "sq1_%s.log" % case)
        contacts = None
        if os.path.exists(log):
            m = re.search(r"contacts (\d+)", open(log).read())
            contacts = int(m.group(1)) if m else None
        entry = {"contacts_per_launch": contacts, "kernels": {}}
        for k, cs in merged.items():
            if not any(k.startswith(p) for p in MAIN):
                continue
            d = dict(cs)
            if "SQ_INSTS_VALU" in cs:
                d["valu_insts_per_launch"] = cs["SQ_INSTS_VALU"]
            if cs.get("SQ_ACTIVE_INST_VALU"):
                d["lane_utilisation"] = cs.get("SQ_THREAD_CYCLES_VALU", 0.0) / (cs["SQ_ACTIVE_INST_VALU"] * 64.0)
            if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
                d["hbm_bytes_per_launch"] = (2.0 * cs.get("FETCH_SIZE", 0.0) + cs.get("WRITE_SIZE", 0.0)) * 1024.0
                if contacts:
                    d["hbm_bytes_per_contact"] = d["hbm_bytes_per_launch"] / contacts
            entry["kernels"][k] = d
        out[case] = entry
    doc = {"command": "tools/profile_r3.sh: rocprofv3 --pmc <counters> --output-format csv -- python3 tools/pmc_case.py <case> "
                      "(SQ pass 1, SQ pass 2, FETCH_SIZE, WRITE_SIZE: four separate counter-only runs per case)",
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; "
                         "both are KB per dispatch",
           "mean_over": "launches after the first (warm-up) of each case",
           "cases": out}
    path = os.path.join(src, "counters.json")
    if cases:
        json.dump(doc, open(path, "w"), indent=1)
    legs = []
    for d in glob.glob(os.path.join(src, "kt_*")):
        if not os.path.isdir(d):
            continue
        leg = os.path.basename(d)[3:]
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(src, "kernel_stats_%s.csv" % leg))
            legs.append(leg)
    if dst:
        os.makedirs(dst, exist_ok=True)
        if cases:
            # merge case by case: a later run of one case replaces only that case
            old = {}
            if os.path.exists(os.path.join(dst, "counters.json")):
                old = json.load(open(os.path.join(dst, "counters.json"))).get("cases", {})
            old.update(out)
            doc["cases"] = old
            json.dump(doc, open(os.path.join(dst, "counters.json"), "w"), indent=1)
        for leg in legs:
            shutil.copy(os.path.join(src, "kernel_stats_%s.csv" % leg), os.path.join(dst, "kernel_stats_%s.csv" % leg))
    brief = {c: {k: {"valu": v.get("valu_insts_per_launch"), "lane_util": round(v.get("lane_utilisation", 0), 3),
                     "hbm_MB": round(v.get("hbm_bytes_per_launch", 0) / 1e6, 2)} for k, v in e["kernels"].items()} for c, e in out.items()}
    print(json.dumps({"cases": brief, "stats_legs": legs}))


if __name__ == "__main__":
    main()
