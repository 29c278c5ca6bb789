#!/bin/bash
# Runs on the GPU box (through gpurun).  Per-leg evidence for bench.py's rooflines, one workload per run so that a
# kernel name maps to ONE problem size:
#   stats:<leg>   rocprofv3 --kernel-trace --stats of `bench.py --legs <leg>` (leg = none | single_pile | c4 | ...)
#   <case>        four counter-only passes over tools/pmc_case.py <case> (tile | quad | c4 | coupled | matvec | c2):
#                 SQ pass 1, SQ pass 2, FETCH_SIZE, WRITE_SIZE -- separate runs, --pmc only (no trace domains)
# then tools/profile_summary.py folds everything under OUTDIR into counters.json + kernel_stats_<leg>.csv.
# usage: tools/profile_r3.sh OUTDIR item...        (the program comes directly after `--`: no env / bash -c hop)
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"
SQ2="SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES"
SQ3="SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum"
for c in "$@"; do
  case "$c" in
    stats:*)
      leg="${c#stats:}"
      timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$leg" -o kt -- python3 "$R/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 --legs "$leg" > "$OUT/kt_$leg.log" 2>&1 || echo "stats $leg failed"
      echo "stats $leg done"
      ;;
    *)
      timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d "$OUT/sq1_$c" -o p -- python3 "$R/tools/pmc_case.py" "$c" > "$OUT/sq1_$c.log" 2>&1 || echo "sq1 $c failed"
      timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d "$OUT/sq2_$c" -o p -- python3 "$R/tools/pmc_case.py" "$c" > "$OUT/sq2_$c.log" 2>&1 || echo "sq2 $c failed"
      timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$c" -o p -- python3 "$R/tools/pmc_case.py" "$c" > "$OUT/fetch_$c.log" 2>&1 || echo "fetch $c failed"
      timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$c" -o p -- python3 "$R/tools/pmc_case.py" "$c" > "$OUT/write_$c.log" 2>&1 || echo "write $c failed"
      echo "counters $c done"
      ;;
  esac
done
python3 "$R/tools/profile_summary.py" "$OUT"
