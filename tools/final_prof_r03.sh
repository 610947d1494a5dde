# final profiles of round 3 (run from the repo root on the GPU box); every step appends to $O/progress.txt so that the run is never silent
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final3
mkdir -p $O
step() { echo "$(date +%T) $*" >> $O/progress.txt; }
step bench; timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
step bench_all; FRIES_BENCH_TOPK=80 timeout -k 10 200 python bench.py --cpu-iters 0 > $O/bench_all_kernels.json 2> /dev/null; echo "bench_all rc=$?"
step rocprof_stats; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-iters 0 > $O/prof_stdout.txt 2>&1; echo "prof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/gap_analysis.py $f > $O/gaps.txt
rm -rf $O/prof
step pmc_fetch; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --cpu-iters 0 --profile-steps 1 > $O/pmc_fetch_stdout.txt 2>&1; echo "pmc fetch rc=$?"
step pmc_write; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --cpu-iters 0 --profile-steps 1 > $O/pmc_write_stdout.txt 2>&1; echo "pmc write rc=$?"
python tests/scripts/gpu_pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_merge.txt 2>&1; echo "merge rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write
step rehearsal; FRIES_BENCH_SHARE_GPU=1 FRIES_BENCH_BACKEND=gloo FRIES_BENCH_TRANSPORT=torch timeout -k 10 300 python bench.py --gpus 2 --cpu-iters 0 --steps 30 --warmup 5 > $O/rehearsal_2ranks.json 2> $O/rehearsal.err; echo "rehearsal rc=$?"
step fks_stats; timeout -k 10 200 python tests/scripts/gpu_fks_stats.py > $O/fks_replay_stats.txt 2>&1; echo "fks stats rc=$?"
step soak; timeout -k 10 300 python tests/scripts/gpu_soak.py 1000000 3000 > $O/soak.txt 2>&1; echo "soak rc=$?"
step facade; timeout -k 10 400 python tests/scripts/gpu_facade_cost.py > $O/facade_cost.txt 2>&1; echo "facade rc=$?"
step hh; timeout -k 10 200 python tests/scripts/gpu_hh_scale.py 12 1000000 -1 10 > $O/hh_scale.txt 2>&1; echo "hh rc=$?"
step done
ls -la $O
tail -1 $O/bench.json | cut -c1-400
