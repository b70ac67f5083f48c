#!/bin/bash
# GPU box: tests/_ddp_gpu_child.py (graph = 1) N times; prints the runs whose plain and data-parallel trainers disagree
N=${1:-40}
for i in $(seq 1 $N); do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + i)) python tests/_ddp_gpu_child.py 1 1e-4 3 2>/dev/null | grep DDPCHILD | python -c "
import json,sys
o=json.loads(sys.stdin.read().split('DDPCHILD ',1)[1])
rel=lambda a,b: abs(a-b)/max(abs(b),1e-30)
w=max(rel(o['losses_ddp'][0][k],o['losses_plain'][0][k]) for k in range(8))
wl=max(rel(o['losses_ddp'][s][k],o['losses_plain'][s][k]) for s in range(3) for k in range(8))
bad = w>1e-5 or o['param_dist_rel']>0.5*o['param_moved_rel'] or not o['views_intact'] or o['stats']['finish']!=0
print('run $i', 'BAD' if bad else 'ok', 'step0', '%.2e'%w, 'all', '%.2e'%wl, 'pdist', '%.3e'%o['param_dist_rel'], 'moved', '%.3e'%o['param_moved_rel'], 'exp_avg_rel', '%.2e'%o['exp_avg_rel'], o['stats'], o['graph_captured'])
"
done
