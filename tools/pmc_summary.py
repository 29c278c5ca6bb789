"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-single
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-single
    python3 tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01/pmc_traffic.json

Counters are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled; WRITE_SIZE is taken as is.
"""
import csv, glob, json, os, re, sys


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    acc = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                m = re.search(r"namespace\)::(\w+)", row["Kernel_Name"])
                name = m.group(1) if m else row["Kernel_Name"].split("(")[0]
                acc.setdefault(name, []).append(float(row["Counter_Value"]))
    return acc


def main():
    dir_f, dir_w, out = sys.argv[1:4]
    contacts = int(sys.argv[4]) if len(sys.argv) > 4 else 24 * 16384
    only = sys.argv[5] if len(sys.argv) > 5 else None     # merge just this kernel's figure into an existing file
    store_as = sys.argv[6] if len(sys.argv) > 6 else only  # ... under this key (one kernel, two cases: "quad_solve_kernel(patches)")
    fetch, write = per_kernel(dir_f, "FETCH_SIZE"), per_kernel(dir_w, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        kernels[name] = {
            "FETCH_SIZE_KB_mean": fm, "dispatches_fetch": len(f),
            "WRITE_SIZE_KB_mean": wm, "dispatches_write": len(w),
            "hbm_bytes_raw": (fm + wm) * 1024.0,
            "hbm_bytes_corrected": (2.0 * fm + wm) * 1024.0,
        }
    doc = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                   "--cpu-seconds 0 --no-single (two separate passes), summarised by tools/pmc_summary.py",
        "workload": "C3 x %d piles per launch = %d contacts, GS 100 sweeps fp64" % (contacts // 16384, contacts),
        "contacts_per_launch": contacts,
        "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
        "kernels": kernels,
    }
    # what bench.py reads for roofline.traffic
    doc["hbm_bytes_per_contact_per_launch"] = {k: v["hbm_bytes_corrected"] / contacts for k, v in kernels.items()
                                               if not k.startswith("__amd")}
    if only:   # one case of tools/pmc_case.py: keep the other kernels' figures of the file
        old = json.load(open(out)) if os.path.exists(out) else {"cases": {}, "hbm_bytes_per_contact_per_launch": {}}
        old.setdefault("cases", {})[store_as] = {"contacts_per_launch": contacts, "kernel": kernels.get(only),
                                                 "bytes_per_contact": doc["hbm_bytes_per_contact_per_launch"].get(only)}
        old["hbm_bytes_per_contact_per_launch"][store_as] = doc["hbm_bytes_per_contact_per_launch"].get(only)
        old["command"] = ("rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/pmc_case.py <case> (two separate "
                          "passes per case), summarised by tools/pmc_summary.py")
        old["correction"] = doc["correction"]
        doc = old
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: round(v["hbm_bytes_corrected"] / 1e6, 2) for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
