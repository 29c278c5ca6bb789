mkdir -p gpurun_out/r3
for cap in 224 128 96 64 48 32 0; do
  echo "cap=$cap"
  if [ "$cap" = "0" ]; then timeout -k 10 200 python tools/gpu_time_wall.py 2>&1 | grep wall; else EGS_PATCH_CAP=$cap timeout -k 10 200 python tools/gpu_time_wall.py 2>&1 | grep wall; fi
done
