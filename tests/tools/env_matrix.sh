#!/bin/bash
# Run the GPU suite under each non-default schedule switch: every path
# must give the oracle's bits, whichever schedule is forced.  One pytest process at a time.
# Usage (on the GPU box): bash tests/tools/env_matrix.sh > gpurun_out/env_matrix.log
for setting in EGS_QUAD=0 EGS_QUAD=1 EGS_ISO=2 EGS_ISO=0 EGS_TILE=512 EGS_TILE=128 EGS_QUAD_TILE=256 EGS_PATCH=0 \
               EGS_QUAD_PATCH=0 EGS_LANE_ORDER=0 EGS_SLOT_BANKS=0 EGS_BROADPHASE=grid EGS_BROADPHASE=pairs EGS_MV_TILE=256 EGS_MV_NT=0 EGS_STEP=0 EGS_STEP=1 "EGS_STEP=1 EGS_QUAD=0" "EGS_STEP=0 EGS_QUAD=1" "EGS_STEP=1 EGS_QUAD=1" EGS_RUNS=0 EGS_RUNS=2 "EGS_LEAN=1 EGS_QUAD=0 EGS_ISO=2" "EGS_LEAN=1 EGS_QUAD=0 EGS_TILE=512" EGS_PATCH_RUNS=0 EGS_GRANULES=0 "EGS_GRANULES=0 EGS_PATCH_RUNS=0" EGS_PATCH_ORDER=0 EGS_PATCH_SHAPE=blobs EGS_PATCH_SHAPE=chunks EGS_PATCH_CAP=128 EGS_DENSE_BORDER=0 EGS_DENSE_GUESS=0 EGS_CHOL_FUSED=0; do
  echo "== $setting"
  # a crashed run (no summary line) ends the matrix: no further GPU work after a fault
  env $setting timeout -k 10 600 python -m pytest tests -m gpu -q > /tmp/env_matrix_one.log 2>&1
  grep -E "passed|failed|FAILED|ERROR" /tmp/env_matrix_one.log | tail -20 || { tail -5 /tmp/env_matrix_one.log; exit 1; }
done
