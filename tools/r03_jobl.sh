#!/bin/bash
# round 3: fused launch on 1024-wide planes: two halo slots (three workgroups per CU) against three (two per CU)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_c5_dd_ab.txt
: > $OUT
run() {
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value %.2f ms/step %.4f  update %.4f (%.3f)  dirdot %.4f (%.3f)" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"]))')" | tee -a $OUT
}
run "c5 slab default (DD=7)      " SCHWZ_DUMMY=1 -- --strong 1024,1024,128
run "c5 slab DD=5 (2 halo slots) " SCHWZ_HIP_LIB=$ROOT/schwarz-lib_amd/lib/libschwz_hip_dd5.so -- --strong 1024,1024,128
run "c5 slab default (DD=7)      " SCHWZ_DUMMY=1 -- --strong 1024,1024,128
run "slab DD=5                   " SCHWZ_HIP_LIB=$ROOT/schwarz-lib_amd/lib/libschwz_hip_dd5.so -- --strong 512,512,64
run "slab default                " SCHWZ_DUMMY=1 -- --strong 512,512,64
