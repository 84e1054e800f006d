#!/bin/bash
# In-box A/B of several builds of libschwz_hip.so (timings differ by several % between boxes):
#   tools/ab.sh rounds lib/a.so lib/b.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    echo "== $L"
    SCHWZ_HIP_LIB=$ROOT/schwarz-lib_amd/$L timeout -k 10 300 python3 $ROOT/tools/spmv_probe.py --variants 0 --reps 20 $PROBE_ARGS 2>&1 | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-40s %.4f ms' % (d['kernel'][:40], d.get('ms_per_cg_iteration', d['ms'])))" || exit 1
  done
done
