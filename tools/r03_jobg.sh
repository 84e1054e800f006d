#!/bin/bash
# round 3: whole GPU suite; setup stages after the threaded builders; fused-launch band height on 1024-wide planes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_jobg.txt
: > $OUT
python -m pytest tests/ -x -q -m gpu > $ROOT/gpurun_out/r03_gputests_g.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT
tail -3 $ROOT/gpurun_out/r03_gputests_g.txt | tee -a $OUT
python3 $ROOT/tools/setup_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT
python3 $ROOT/tools/setup_probe.py --shape 1024,1024,128 2>&1 | grep -v amdgpu.ids | tee -a $OUT
run() {
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms frac %.3f  dirdot %.4f ms frac %.3f  flav %d setup %.2f s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"], d["config"]["cg_flavour"], d["setup_s"]))')" | tee -a $OUT
}
run "c5/4 TDIR2048 (default)" SCHWZ_DUMMY=1 -- --strong 1024,1024,32
run "c5/4 TDIR1024          " SCHWZ_SWEEP_TDIR=1024 -- --strong 1024,1024,32
run "c5 slab TDIR2048       " SCHWZ_DUMMY=1 -- --strong 1024,1024,128
run "c5 slab TDIR1024       " SCHWZ_SWEEP_TDIR=1024 -- --strong 1024,1024,128
run "cube                   " SCHWZ_DUMMY=1 --
