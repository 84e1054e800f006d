#!/bin/bash
# segment length of the fused launch's own table (SCHWZ_SWEEP_LDIR) on the cube
for rep in 1 2; do for l in auto 16 24 32 48 64; do
  if [ $l = auto ]; then e="SCHWZ_X=0"; else e="SCHWZ_SWEEP_LDIR=$l"; fi
  echo "L_dir=$l: $(env $e python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ttr --no-plain-loop 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms/step  upd %.4f ms  dirdot %.4f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_spmv"]["avg_launch_ms"]))')"
done; done
