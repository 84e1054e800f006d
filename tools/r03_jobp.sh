#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_gen_walk.txt
: > $OUT
SCHWZ_FUZZ_SEEDS=24 python -m pytest tests/test_gpu_kernels.py -x -q -k "z_sweep_walk_on" > $ROOT/gpurun_out/r03_gputests_p.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT
tail -4 $ROOT/gpurun_out/r03_gputests_p.txt | tee -a $OUT
run() {
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid= "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value %.1f ms/step %.4f  update %.4f (%.3f)  dirdot %.4f (%.3f) flav %d reduction %.17g" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"], d["config"]["cg_flavour"], d["residual_reduction_in_timed_steps"]))')" | tee -a $OUT
}
run "200 gen off " SCHWZ_SWEEP_GEN=0 -- --size 200
run "200 gen on  " SCHWZ_DUMMY=1 -- --size 200
run "300 gen off " SCHWZ_SWEEP_GEN=0 -- --size 300
run "300 gen on  " SCHWZ_DUMMY=1 -- --size 300
run "cube        " SCHWZ_DUMMY=1 --
