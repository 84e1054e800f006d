#!/bin/bash
# round 3: lazy last iteration + restriction inside the solver: tests, then in-box A/B of the bench line
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_lazy_ab.txt
: > $OUT
python -m pytest tests/test_gpu_ras.py tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q > $ROOT/gpurun_out/r03_gputests_i.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT
tail -3 $ROOT/gpurun_out/r03_gputests_i.txt | tee -a $OUT
run() {
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 30 --warmup 3 --no-ttr --no-cpu-baseline --no-mirror --no-shapes --strong-grid "" "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value %.1f ms/step %.4f  update %.4f  dirdot %.4f  plain_loop %s reduction %.17g" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_spmv"]["avg_launch_ms"], d.get("csr_plain_loop",{}).get("value"), d["residual_reduction_in_timed_steps"]))')" | tee -a $OUT
}
for rep in 1 2; do
run "cube eager            " SCHWZ_CG_LAZYLAST=0 SCHWZ_RESTRICT_FUSE=0 --
run "cube lazy only        " SCHWZ_CG_LAZYLAST=1 SCHWZ_RESTRICT_FUSE=0 --
run "cube lazy + restrict  " SCHWZ_DUMMY=1 --
done
run "slab eager            " SCHWZ_CG_LAZYLAST=0 SCHWZ_RESTRICT_FUSE=0 -- --strong 512,512,64 --no-plain-loop
run "slab lazy + restrict  " SCHWZ_DUMMY=1 -- --strong 512,512,64 --no-plain-loop
