#!/bin/bash
# round 5, GPU call 30: longer runs of the N > 1 line on one device (gloo): 2 ranks x 300 steps row-sharded, 4 ranks x 100 steps replicated
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_30
mkdir -p $O
export CMDIAD_BENCH_ONE_DEVICE=1 OMP_NUM_THREADS=2
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 300 --warmup 4 --bank sharded --no-extras > $O/n2.json 2> $O/n2.err; echo "n2 rc=$?" | tee -a $O/rc.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 4 --steps 100 --warmup 4 --no-extras > $O/n4.json 2> $O/n4.err; echo "n4 rc=$?" | tee -a $O/rc.log
for f in n2 n4; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['bank'][:40], d.get('rehearsal','')[:30])"; done
