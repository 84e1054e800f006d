#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_stream_order.txt
: > $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -k "stream or spmv" > $ROOT/gpurun_out/r03_gputests_r.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT
tail -3 $ROOT/gpurun_out/r03_gputests_r.txt | tee -a $OUT
for ord in 1 0 1 0; do
  echo "SCHWZ_STREAM_ORDER=$ord" | tee -a $OUT
  SCHWZ_STREAM_ORDER=$ord python3 -c "
import os, sys
sys.path.insert(0, '$ROOT/schwarz-lib_amd'); sys.path.insert(0, '$ROOT/tools')
for k in ('SCHWZ_SPMV_PAIR','SCHWZ_SPMV_PATTERN','SCHWZ_SPMV_DICT'): os.environ[k]='0'
import torch, schwz_amd as S
from spmv_probe import timeit
for shape in ((1024,1024,32),(1024,1024,128)):
    prob=S.Problem.laplacian(3,*shape); sd=S.Subdomain(prob,1,0,2,S.partition_regular(prob.N,1)); rp,col,val=sd.local_matrix(); A=S.Csr(rp,col,val); n=len(rp)-1
    del rp,col,val
    x=torch.randn(n,dtype=torch.float64,device='cuda'); y=torch.zeros(n,dtype=torch.float64,device='cuda'); st=torch.cuda.current_stream().cuda_stream
    ms=[round(timeit(torch, lambda: A.spmv(x.data_ptr(), y.data_ptr(), 1.0, 0.0, 6, st), 20),4) for _ in range(2)]
    print(shape, ms, 'frac %.3f' % (A.algorithmic_bytes()/min(ms)/1e6/8000), flush=True)
    del A,x,y,sd,prob; torch.cuda.empty_cache()
" 2>&1 | grep -v amdgpu | tee -a $OUT
done
