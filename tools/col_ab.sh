# the start-up regime at m = 1e6: the chain checked against its element-by-element form, then timed in both forms
FRIES_FSQ_CHECK=1 FRIES_DBG=1 timeout -k 10 200 python tests/scripts/gpu_collapse_time.py 1000000 12 > gpurun_out/col_check.log 2>&1 || { tail -n 5 gpurun_out/col_check.log; exit 1; }
echo "check: $(grep 'total ms' gpurun_out/col_check.log)"; grep "fries\]" gpurun_out/col_check.log | cut -c1-300
for mp in 1 0; do
  FRIES_FSQ_MAPS=$mp COLLAPSE_PROF=1 FRIES_DBG=1 timeout -k 10 120 python tests/scripts/gpu_collapse_time.py 1000000 12 > gpurun_out/col_maps$mp.log 2>&1 || exit 1
  echo "maps $mp: $(grep k_fsq_chain gpurun_out/col_maps$mp.log) $(grep 'total ms' gpurun_out/col_maps$mp.log)"
done
