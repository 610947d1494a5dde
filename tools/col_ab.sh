for sm in 32 0 64 128; do
  FRIES_FSQ_SPARSE_MAX=$sm COLLAPSE_PROF=1 timeout -k 10 120 python tests/gpu_collapse_time.py 1000000 12 > gpurun_out/col_sm$sm.log 2>&1 || exit 1
  echo "sparse_max $sm: $(grep k_fsq_chain gpurun_out/col_sm$sm.log) $(grep 'total ms' gpurun_out/col_sm$sm.log)"
done
