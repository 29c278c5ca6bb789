#!/bin/bash
# SQ counters of one tools/pmc_case.py case (one pass, counters only).  usage: tools/gpu_pmc_sq.sh OUTDIR case
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq_$2" -o p -- python3 "$R/tools/pmc_case.py" "$2" > "$OUT/sq_$2.log" 2>&1 || echo "sq $2 failed"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$OUT/sq2_$2" -o p -- python3 "$R/tools/pmc_case.py" "$2" > "$OUT/sq2_$2.log" 2>&1 || echo "sq2 $2 failed"
python3 - "$OUT" "$2" <<'PY'
import csv, glob, sys, re, json, os
out, case = sys.argv[1], sys.argv[2]
acc = {}
for d in ("sq_" + case, "sq2_" + case):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"namespace\)::(\w+)", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"].split("(")[0]
            if "solve" not in name and "matvec" not in name: continue
            acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open(os.path.join(out, "sq_%s.json" % case), "w"), indent=1)
print(json.dumps(res))
PY
