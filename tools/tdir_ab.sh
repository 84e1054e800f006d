#!/bin/bash
# in-box A/B of the band height of the fused direction launch (SCHWZ_SWEEP_TDIR) on the cube, the 512 x 512 x 64
# slab and (optionally) the 1024 x 1024 x 128 slab of configs[4]
set -o pipefail
run() {  # label, env, bench args
  echo "$1: $(env $2 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ttr --no-plain-loop $3 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms/step  upd %.4f ms  dirdot %.4f ms (%.3f)" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"]))')"
}
for rep in 1 2; do
  run "cube  T_dir=512 (= update launch)" "SCHWZ_SWEEP_TDIR=512" ""
  run "cube  T_dir=1024 (default)       " "SCHWZ_SWEEP_TDIR=1024" ""
  run "cube  T_dir=2048                 " "SCHWZ_SWEEP_TDIR=2048" ""
  run "slab  T_dir=1024 (= update, default)" "SCHWZ_SWEEP_TDIR=1024" "--strong 512,512,64"
  run "slab  T_dir=2048                    " "SCHWZ_SWEEP_TDIR=2048" "--strong 512,512,64"
done
if [ "$1" = "big" ]; then
  run "1024^2x128 T_dir=1024 (= update)" "SCHWZ_SWEEP_TDIR=1024" "--strong 1024,1024,128 --steps 8"
  run "1024^2x128 T_dir=2048 (default) " "SCHWZ_SWEEP_TDIR=2048" "--strong 1024,1024,128 --steps 8"
fi
