#!/bin/bash
# round 3: one-line bands (halo twice the own rows) with the three-ahead halo schedule; H3 on other line lengths; setup timing
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_h3_ab2.txt
: > $OUT
run() {
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms frac %.3f  dirdot %.4f ms frac %.3f  flav %d" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"], d["config"]["cg_flavour"]))')" | tee -a $OUT
}
run "slab T1024 (default)      " SCHWZ_SWEEP_H3=0 -- --strong 512,512,64
run "slab T512 H3=0            " SCHWZ_SWEEP_T=512 SCHWZ_SWEEP_H3=0 -- --strong 512,512,64
run "slab T512 H3=1            " SCHWZ_SWEEP_T=512 SCHWZ_SWEEP_H3=1 -- --strong 512,512,64
run "slab T512 H3=1 TDIR512    " SCHWZ_SWEEP_T=512 SCHWZ_SWEEP_H3=1 SCHWZ_SWEEP_TDIR=512 -- --strong 512,512,64
run "slab T512 H3=0 TDIR512    " SCHWZ_SWEEP_T=512 SCHWZ_SWEEP_H3=0 SCHWZ_SWEEP_TDIR=512 -- --strong 512,512,64
run "320 T512 H3=0             " SCHWZ_SWEEP_H3=0 -- --size 320
run "320 T512 H3=1             " SCHWZ_SWEEP_H3=1 -- --size 320
run "320 T1024 H3=1            " SCHWZ_SWEEP_H3=1 SCHWZ_SWEEP_T=1024 -- --size 320
run "192 H3=0                  " SCHWZ_SWEEP_H3=0 -- --size 192
run "192 H3=1                  " SCHWZ_SWEEP_H3=1 -- --size 192
run "384 H3=0                  " SCHWZ_SWEEP_H3=0 -- --size 384
run "384 H3=1                  " SCHWZ_SWEEP_H3=1 -- --size 384
python3 $ROOT/tools/setup_probe.py > $ROOT/gpurun_out/r03_setup_probe.txt 2>&1
python3 $ROOT/tools/setup_probe.py --shape 512,512,64 >> $ROOT/gpurun_out/r03_setup_probe.txt 2>&1
grep -v amdgpu.ids $ROOT/gpurun_out/r03_setup_probe.txt
