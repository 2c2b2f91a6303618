"""configs[3] shard shapes on one GPU: prints bench.py's `fake_world` leg (per class and world size: the per-rank distance GEMM).
    python tools/l2_segments.py [classes] [worlds]         e.g.  bagel,peach 1,2,4,8
CMDIAD_L2_SEG_SPLITS=<n> forces the library ranges per query tile of the segments launch (default: csrc/l2min.hip pick_splits)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

classes = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("bagel", "peach")
worlds = tuple(int(w) for w in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)
leg = bench.fake_world_leg(torch.device("cuda", 0), classes, worlds)
for s in leg["shapes"]:
    print(json.dumps(s))
