#!/bin/bash
# where does the z-sweep walk start to pay?  one subdomain of 256 x 256 x nz, walk off / on
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_sizes.txt
: > $OUT
for shape in 256,256,16 256,256,32 256,256,64 256,256,128 512,512,16 512,512,32; do
  for sw in 0 1; do
    line=$(SCHWZ_CG_SWEEP=$sw SCHWZ_SPMV_SWEEP=2 python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --strong $shape 2>/dev/null)
    echo "$shape sweep=$sw $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms  dot/dirdot %.4f ms launches/iter %d" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_spmv"]["avg_launch_ms"], d["config"]["cg_launches_per_iteration"]))')" | tee -a $OUT
  done
done
