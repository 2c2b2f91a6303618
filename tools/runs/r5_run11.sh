#!/bin/bash
# round 5, GPU call 11: two real ranks, diagnostics of the pipeline difference
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_11
mkdir -p $O
WORLD2_DIAG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 tests/world2_worker.py > $O/w2.log 2>&1; echo "world2 rc=$?" | tee -a $O/rc.log
grep -h "diag\|Assertion\|world2" $O/w2.log | tail -n 20
