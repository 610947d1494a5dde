# (the variant libraries live in scratch/ab/, which is git-ignored but travels to the GPU box)
# A/B timing on one box: bench.py with each library in scratch/ab/*.so and the default build, alternating, twice
set -o pipefail
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  lib=${v%%:*}; envs=""; if [ "$lib" != "$v" ]; then envs=${v#*:}; fi; tag=$(echo $v | tr ':=,' '___')
  if [ "$lib" = "default" ]; then unset FRIES_LIB; else export FRIES_LIB=$GRAFT_REPO_ROOT/scratch/ab/$lib.so; fi
  env $(echo $envs | tr ',' ' ') FRIES_BENCH_TOPK=80 timeout -k 10 300 python bench.py --cpu-iters 0 --steps 100 > gpurun_out/ab_${tag}_$rep.json 2> gpurun_out/ab_${tag}_$rep.err || { echo "bench $v failed"; tail -5 gpurun_out/ab_${tag}_$rep.err; exit 1; }
  python - $tag $rep <<PY
import json, sys
v, rep = sys.argv[1], sys.argv[2]
d = json.loads(open(f"gpurun_out/ab_{v}_{rep}.json").read().strip().splitlines()[-1])
tk = d["top_kernels"]
want = ["k_fks_sweep", "k_fks_sweep_rec", "k_fks_sweep_light", "k_fks_close", "k_sys_count", "k_sys_write", "k_prep", "k_fks_scan", "k_fks_totals", "k_seq_chain", "k_spawn_lookup", "k_final_eval", "k_death_clone", "k_seg_sum"]
print(f"{v:8s} {rep} {d['value']:7.2f} it/s  ktime {d['kernel_time_ms_per_iter']:.3f} | " + " ".join(f"{k[2:]}={tk[k]['ms_per_iter']*1000/max(tk[k]['calls_per_iter'],1e-9):.1f}" for k in want if k in tk))
PY
done
done
