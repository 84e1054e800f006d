#!/bin/bash
# round 3: rehearsal of the N > 1 bench line on the 1-GPU box (ranks share the GPU, halos over gloo): the launch
# path, the weak + strong legs and the wall time of the whole command
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for N in 2 4; do
    t0=$(date +%s.%N)
    python3 $ROOT/bench.py --gpus $N > $ROOT/gpurun_out/r03_rehearsal_gpus$N.json 2> $ROOT/gpurun_out/r03_rehearsal_gpus$N.err
    rc=$?
    t1=$(date +%s.%N)
    echo "N=$N rc=$rc wall $(python3 -c "print('%.1f' % ($t1 - $t0))") s"
    python3 -c "
import json,sys
d=json.loads(open('$ROOT/gpurun_out/r03_rehearsal_gpus$N.json').read())
print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['config']['exchange_backend'], d.get('strong'), d.get('time_to_residual_iters'))
" || tail -5 $ROOT/gpurun_out/r03_rehearsal_gpus$N.err
done
