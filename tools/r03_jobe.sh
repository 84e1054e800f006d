#!/bin/bash
# round 3: halo schedule of the update walk (SCHWZ_SWEEP_H3), 16-run records (192^3), band height at 320^3; tests first
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/r03_h3_ab.txt
: > $OUT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ras.py tests/test_gpu_bench_ras.py -x -q > $ROOT/gpurun_out/r03_gputests_e.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT
tail -3 $ROOT/gpurun_out/r03_gputests_e.txt | tee -a $OUT
run() {  # label, env..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    line=$(env "${envs[@]}" python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" "$@" 2>/dev/null)
    echo "$label $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms frac %.3f  dirdot %.4f ms frac %.3f  flav %d  reduction %.17g" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"], d["config"]["cg_flavour"], d["residual_reduction_in_timed_steps"]))')" | tee -a $OUT
}
for rep in 1 2; do
run "cube H3=0   " SCHWZ_SWEEP_H3=0 --
run "cube H3=1   " SCHWZ_SWEEP_H3=1 --
run "cube auto   " SCHWZ_DUMMY=1 --
run "slab H3=0   " SCHWZ_SWEEP_H3=0 -- --strong 512,512,64
run "slab H3=1   " SCHWZ_SWEEP_H3=1 -- --strong 512,512,64
run "slab auto   " SCHWZ_DUMMY=1 -- --strong 512,512,64
done
run "c5/4 H3=0   " SCHWZ_SWEEP_H3=0 -- --strong 1024,1024,32
run "c5/4 H3=1   " SCHWZ_SWEEP_H3=1 -- --strong 1024,1024,32
run "c5/4 H3=0 T2048 " SCHWZ_SWEEP_H3=0 SCHWZ_SWEEP_TDIR=2048 -- --strong 1024,1024,32
run "192 rle8    " SCHWZ_SPMV_RLE=8 -- --size 192
run "192 rle16   " SCHWZ_DUMMY=1 -- --size 192
run "192 rle16 T1024" SCHWZ_SWEEP_T=1024 -- --size 192
run "320 T512    " SCHWZ_DUMMY=1 -- --size 320
run "320 T1024   " SCHWZ_SWEEP_T=1024 -- --size 320
run "320 T1024 TDIR2048" SCHWZ_SWEEP_T=1024 SCHWZ_SWEEP_TDIR=2048 -- --size 320
run "200 default " SCHWZ_DUMMY=1 -- --size 200
