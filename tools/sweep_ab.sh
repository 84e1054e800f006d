#!/bin/bash
# In-box A/B of the z-sweep walk of the CG update launch: bench lines (ms/step, update-launch time)
# for the cube and the slab with the walk off / on and a few band / segment settings.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_ab.txt
: > $OUT
run() {  # label, env..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms frac %.3f  dot/dirdot %.4f ms frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"]))')" | tee -a $OUT
}
for shape in cube slab; do
    if [ $shape = cube ]; then ARGS=""; else ARGS="--strong 512,512,64"; fi
    run "$shape sweep-off          " SCHWZ_CG_SWEEP=0 -- $ARGS
    run "$shape sweep default (fused)   " -- $ARGS
    run "$shape sweep, three launches   " SCHWZ_CG_FUSEDIR=0 -- $ARGS
    for T in 512 1024; do
        for L in 24 32 48; do
            run "$shape sweep T=$T L=$L   " SCHWZ_SWEEP_T=$T SCHWZ_SWEEP_L=$L -- $ARGS
        done
    done
    run "$shape sweep-off again    " SCHWZ_CG_SWEEP=0 -- $ARGS
done
