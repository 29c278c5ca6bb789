mkdir -p gpurun_out/r3
rm -f gpurun_out/r3/lean_sweep.log
for n in 8 16 32; do
  EGS_TILE=512 EGS_LEAN=2 timeout -k 10 120 python tools/gpu_time_batch.py $n 20 >> gpurun_out/r3/lean_sweep.log 2>&1 || echo "lean512 $n failed" >> gpurun_out/r3/lean_sweep.log
done
for n in 8 24 32; do
  EGS_ISO=2 EGS_LEAN=2 timeout -k 10 120 python tools/gpu_time_batch.py $n 20 >> gpurun_out/r3/lean_sweep.log 2>&1 || echo "lean256 $n failed" >> gpurun_out/r3/lean_sweep.log
done
cat gpurun_out/r3/lean_sweep.log
