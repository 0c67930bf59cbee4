set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python bench.py > $O/final_c2.json 2> $O/final_c2.err
timeout -k 10 600 python bench.py --workload C4 --candidates 1000000 --queries 1000 --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0 > $O/final_c4.json 2> $O/final_c4.err
timeout -k 10 600 python bench.py --workload C3 --gpus 1 --candidates 12500032 --queries 2000 --exchange rccl --no-cpu-baseline > $O/final_c3.json 2> $O/final_c3.err
python - <<PY
import json
for t in ("c2","c4","c3"):
    d=json.loads(open("$O/final_%s.json"%t).read().strip().splitlines()[-1])
    r=d["roofline"]; h=d["roofline_hbm"]
    print(t, "step %.3f ms value %.3e frac %s occ %s alg %s hbm %s equal %s sync %s" % (d["ms_per_step"], d["value"], r.get("frac"), r.get("frac_at_kernel_occupancy"), r.get("algorithm_instr_frac"), h.get("frac"), d.get("results_equal"), (d.get("synchronous_calls") or {}).get("ms_per_step")))
PY
