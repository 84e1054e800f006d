#!/bin/bash
# round 3: segment length of the walk with the round-3 kernels (short segments in dispatch order?)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_seglen_ab.txt
: > $OUT
run() {
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 30 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value %.1f ms/step %.4f  update %.4f (%.3f)  dirdot %.4f (%.3f)" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"]))')" | tee -a $OUT
}
run "cube default           " SCHWZ_DUMMY=1 --
for L in 8 12 16 24 64; do
run "cube L=$L LDIR=$L        " SCHWZ_SWEEP_L=$L SCHWZ_SWEEP_LDIR=$L --
done
run "cube default           " SCHWZ_DUMMY=1 --
run "slab default           " SCHWZ_DUMMY=1 -- --strong 512,512,64
for L in 8 16 32; do
run "slab L=$L LDIR=$L        " SCHWZ_SWEEP_L=$L SCHWZ_SWEEP_LDIR=$L -- --strong 512,512,64
done
