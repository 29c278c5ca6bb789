#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the bench command and the PMC
# passes (separate runs, counters only) of tools/pmc_case.py.  Usage: tools/gpu_profile.sh OUTDIR case...
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  case "$c" in
    stats)
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$R/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 > "$OUT/kt.log" 2>&1 || echo "stats failed"
      ;;
    *)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${ctr}_$c" -o p -- python3 "$R/tools/pmc_case.py" "$c" > "$OUT/pmc_${ctr}_$c.log" 2>&1 || echo "pmc $ctr $c failed"
      done
      ;;
  esac
done
ls "$OUT"
