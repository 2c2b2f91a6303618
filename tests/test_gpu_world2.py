"""GPU: the row-sharded paths with TWO REAL RANKS.  Every other multi-rank test of this suite is either a "fake world" (W shards
searched in turn by one process), an RCCL world of ONE rank, or a gloo world on the CPU with a torch stand-in for the kernels: none
runs the HIP kernels inside a process group of more than one rank.  Here two processes share the box's one GPU (RCCL refuses two
ranks on one device, so the collectives travel over gloo; the kernels, the shard arithmetic, the per-rank queries, the counts
exchange, the MIN / MAX / SUM reductions and the pipeline's repeat-on-overflow are the real ones): tests/world2_worker.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_sharded_paths_with_two_real_ranks_on_one_gpu():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("OMP_NUM_THREADS", "8")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "tests", "world2_worker.py")],
                         capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["world2"] is True and rec["ranks"] == 2
    assert [c.split("(")[0] for c in rec["checked"]] == ["sharded_search", "sharded_fp32_scoring", "sharded_coreset", "sharded_pipeline"]
