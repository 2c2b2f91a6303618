"""CPU (no GPU): the C-ABI library loads and exports every symbol the header declares; host-side logic of the
drop-in (state_dict names, seeded init parity with the reference, AU-PRO, LR schedule, sharding arithmetic);
the multi-GPU merge path with world_size-2 gloo."""
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    from cmdiad_amd import _native as nat
    L = nat.lib()
    assert L.cmdiad_abi_version() == 6
    hdr = open(os.path.join(REPO, "include", "cmdiad_hip.h")).read()
    declared = set(re.findall(r"\b(cmdiad_[a-z0-9_]+)\s*\(", hdr))
    bound = set(nat.SIGNATURES) | set(nat.SIZE_QUERIES) | {"cmdiad_last_error", "cmdiad_abi_version", "cmdiad_has_ab_variants"}
    assert L.cmdiad_has_ab_variants() == 0   # the production library carries ONE formulation of every kernel
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(L, name), name
    # argument validation works without a GPU (no launch happens on an error path)
    rc = L.cmdiad_fps(None, None, 1, 10, 4, None, None, None, 0, None)
    assert rc == -1 and b"null pointer" in L.cmdiad_last_error()
    assert L.cmdiad_fps_workspace_bytes(2, 24576) == 0 and L.cmdiad_fps_workspace_bytes(2, 50176) == 2 * 50176 * 4


def test_production_kernels_do_not_spill():
    """tools/scratch_report.py on the built library (the AMDGPU metadata of every code object in it; nothing is compiled): no kernel
    of the production library uses scratch, except the two known ones with a dozen bytes outside their loops.  A spill that creeps
    into a hot kernel -- a lambda that keeps an index array addressable, an epilogue that outgrows the register file -- costs
    silently: this makes it a test failure (round 4: both happened on the way, profiles/r4_notes.md sections 11 and 13)."""
    import importlib.util
    from cmdiad_amd import _native as nat
    nat.lib()
    spec = importlib.util.spec_from_file_location("scratch_report", os.path.join(REPO, "tools", "scratch_report.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    ks = sr.kernels(os.path.join(REPO, "cmdiad_amd", "libcmdiad_hip.so"))
    assert len(ks) > 150, len(ks)
    allowed = {"fps_ragged_kernel": 16, "encoder_tail_persist_kernelILb0E": 12}
    bad = {}
    for name, v in ks.items():
        sc = v.get("private_segment_fixed_size", 0)
        if sc and not any(tag in name and sc <= lim for tag, lim in allowed.items()):
            bad[name] = sc
    assert not bad, bad


def test_product_has_no_cpu_fallback_and_no_oracle_import():
    from cmdiad_amd import ops
    with pytest.raises(Exception, match="GPU|cuda|CUDA"):
        ops.fps(torch.zeros(1, 100, 3), 8)
    for root, _, files in os.walk(os.path.join(REPO, "cmdiad_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
    # bench.py: only the cpu_baseline leg (bench_legs/cpu.py) may touch the checker
    import glob
    for f in [os.path.join(REPO, "bench.py")] + glob.glob(os.path.join(REPO, "bench_legs", "*.py")):
        if os.path.basename(f) != "cpu.py":
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(f).read(), re.M), f"{f} imports the oracle"


def test_seeded_init_matches_reference_checksums(golden):
    """Same construction order => same RNG consumption => a seeded default init reproduces the reference's
    (tests/golden/make_golden.py G8)."""
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.models.models import PointTransformer
    g = golden("g8_init.npz")
    torch.manual_seed(123)
    sd = PointTransformer(group_size=128, num_group=1024).state_dict()
    ref_keys = {k[3:] for k in g.files if k.startswith("pm/")}
    assert set(sd) == ref_keys
    for k, v in sd.items():
        np.testing.assert_allclose([v.double().sum().item(), v.double().abs().sum().item()], g["pm/" + k], rtol=1e-9, atol=1e-9)
    torch.manual_seed(321)
    sd = HallucinationCrossModalityNetwork(None, 768, 768).state_dict()
    assert set(sd) == {k[3:] for k in g.files if k.startswith("hn/")}
    for k, v in sd.items():
        np.testing.assert_allclose([v.double().sum().item(), v.double().abs().sum().item()], g["hn/" + k], rtol=1e-9, atol=1e-9)


def test_distillation_head_modules_match_reference_names_and_seeded_init(golden):
    """Conv / FtoI / HRNet heads: same state_dict keys as the reference's modules and the same values after a
    seeded default construction (same construction order => same RNG consumption; tests/golden/make_golden.py G10)."""
    from cmdiad_amd.models import hallucination_network as hn
    from cmdiad_amd.models.hrnet import HRNet
    g = golden("g10_heads.npz")
    build = {"conv_ftof": lambda: hn.HallucinationCrossModalityConv(None, 768, 768),
             "ftoi_mlp": lambda: hn.HallucinationRGBFeatureToXYZInputMLP(types.SimpleNamespace(estimate_depth=False), 768),
             "ftoi_conv": lambda: hn.HallucinationFeatureToInputConv(None, 768),
             "hrnet": lambda: HRNet(512, 768, 0.1)}
    for kind, make in build.items():
        torch.manual_seed(777)
        sd = make().state_dict()
        ref = {k[len(f"init/{kind}/"):]: g[k] for k in g.files if k.startswith(f"init/{kind}/")}
        assert set(sd) == set(ref), kind
        for k, v in sd.items():
            np.testing.assert_allclose([v.double().sum().item(), v.double().abs().sum().item()], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{kind}/{k}")
    with pytest.raises(RuntimeError, match="GPU"):
        build["ftoi_conv"]().hallucination_generation(torch.zeros(1, 3136, 768))


def test_vit_state_dict_uses_timm_names():
    from cmdiad_amd.models.models import VisionTransformer
    from oracle import nets
    assert set(VisionTransformer().state_dict()) == set(nets.synth_state_dict("vit", 0))


def test_au_pro_matches_reference(golden):
    from cmdiad_amd.utils.au_pro_util import calculate_au_pro
    g = golden("g7_aupro.npz")
    gts, preds = list(g["gts"]), list(g["preds"])
    a03, _ = calculate_au_pro(gts, preds)
    a001, _ = calculate_au_pro(gts, preds, 0.01)
    # default = the reference's 100-threshold sampling: the same number, not an approximation of it
    assert abs(a03 - float(g["au_pro_03"])) < 1e-12 and abs(a001 - float(g["au_pro_001"])) < 1e-12
    e03, _ = calculate_au_pro(gts, preds, num_thresholds=None)   # exact curve: close to, not equal to, the sampled value
    assert abs(e03 - float(g["au_pro_03"])) < 2e-3


def test_lr_schedule_and_shard_ranges():
    from cmdiad_amd import engine as eng
    from cmdiad_amd.utils import lr_sched
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    a = types.SimpleNamespace(lr=1e-3, warmup_epochs=10)
    assert lr_sched.adjust_learning_rate(opt, 2.5, a) == pytest.approx(2.5e-4)
    assert opt.param_groups[1]["lr"] == pytest.approx(1.25e-4)
    assert lr_sched.adjust_learning_rate(opt, 12, a) == 1e-3
    for n in (76518, 19129, 100, 113209):
        for w in (1, 2, 4, 8):
            spans = [eng.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert all(lo % 128 == 0 for lo, hi in spans if hi > lo)


def test_unorganize_host_helper(golden):
    from cmdiad_amd.feature_extractors.multiple_features import organized_pc_to_unorganized_pc_no_zeros
    from cmdiad_amd.synth import synth_cloud
    g1 = golden("g1b_unorganize.npz")
    pc, nz = organized_pc_to_unorganized_pc_no_zeros((None, synth_cloud(int(g1["seed"]), float(g1["frac"]))))
    assert pc.shape[2] == int(g1["n"]) and nz.sum() == g1["nz_sum"]


def test_feature_ring_reproduces_dataloader_order(tmp_path):
    """cmdiad_amd.dataset.FeatureRing vs torch's DataLoader over the reference's PreTrainTensorDataset protocol
    (dataset.py:247-265, hallucination_network_pretrain.py:216-225): same batches in the same order for two epochs and
    the same global-RNG state afterwards, for shuffle / drop_last combinations."""
    from torch.utils.data import DataLoader, Dataset
    from cmdiad_amd.dataset import FeatureRing
    for i in range(11):
        torch.save(torch.full((4, 6), float(i)), tmp_path / f"bagel{i}.pt")

    class Files(Dataset):  # the reference's dataset, minus map_location='cuda'
        def __init__(self, root):
            self.root, self.paths = root, os.listdir(root)

        def __len__(self):
            return len(self.paths)

        def __getitem__(self, i):
            return torch.load(os.path.join(self.root, self.paths[i])), 0

    for shuffle, drop in ((True, True), (True, False), (False, False)):
        torch.manual_seed(3407)
        dl = DataLoader(Files(str(tmp_path)), shuffle=shuffle, batch_size=4, drop_last=drop)
        want = [[b[0] for b in dl] for _ in range(2)]
        rng_after = torch.rand(1).item()
        torch.manual_seed(3407)
        ring = FeatureRing(str(tmp_path), 4, shuffle=shuffle, drop_last=drop, device="cpu", depth=3, readers=2)
        got = [[x.clone() for x, _ in ring] for _ in range(2)]
        assert len(ring) == len(want[0])
        assert all(torch.equal(a, b) for e in range(2) for a, b in zip(want[e], got[e])) and len(got[1]) == len(want[1])
        assert rng_after == torch.rand(1).item()



def _free_port():
    """A TCP port nobody listens on right now (the hard-coded rendezvous ports of earlier rounds collide on a shared box)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_gloo(script_text, tmp_path, world=2, timeout=240, extra_env=None):
    """Starts `world` ranks of a worker script (one process each, gloo rendezvous on 127.0.0.1 and a free port), waits for all of
    them and returns their outputs; asserts that every rank exited with 0."""
    script = tmp_path / "w.py"
    script.write_text(script_text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=timeout)[0].decode())
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    assert all(pr.returncode == 0 for pr in procs), outs
    return outs


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch, torch.distributed as td
from cmdiad_amd import engine as eng
from oracle import kernels as ok
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
td.init_process_group("gloo", rank=rank, world_size=world)
g = torch.Generator().manual_seed(5)
bank = torch.randn(1000, 32, generator=g)
bank[700] = bank[100]                      # exact duplicate rows in DIFFERENT shards: tie -> lowest global row
q_all = [bank[torch.randint(0, 1000, (50,), generator=g)] + 0.05 * torch.randn(50, 32, generator=g) for _ in range(world)]
q_all[0][0] = bank[100]
# queries of every rank (all-gather), as engine.gather_queries does for the bf16 operands
mine = q_all[rank].to(torch.bfloat16)
got_q, got_s = eng.gather_queries(mine, mine.float().pow(2).sum(1), td.group.WORLD)
assert torch.equal(got_q.float(), torch.cat(q_all).to(torch.bfloat16).float())
# per-shard search by the CPU oracle, packed exactly like cmdiad_l2_min_keys
lo, hi = eng.shard_range(1000, rank, world)
qs = torch.cat(q_all)
d2 = ((qs[:, None, :].double() - bank[None, lo:hi].double()) ** 2).sum(-1).float()
mv, mi = d2.min(1)
keys = (mv.view(torch.int32).to(torch.int64) << 32) | (mi + lo)
keys = eng.merge_shard_keys(keys, td.group.WORLD)
d2f = ((qs[:, None, :].double() - bank[None].double()) ** 2).sum(-1).float()
rv, ri = d2f.min(1)
assert torch.equal(keys & 0xFFFFFFFF, ri), "merged argmin differs from the single-bank argmin"
assert int(keys[0] & 0xFFFFFFFF) == 100          # the duplicate at row 700 must lose to row 100
assert torch.equal((keys >> 32).to(torch.int32).view(torch.float32), rv)
# best + runner-up planes (ops.new_keys(runner=True)): the runner-up is the nearest row OUTSIDE the winner's group of 16 rows,
# group(row) = (row >> 6, (row >> 2) & 3) on global rows; two MIN all-reduces give what one device searching everything returns
def top2(d2m, first_row):
    rows = torch.arange(d2m.shape[1]) + first_row
    k = (d2m.contiguous().view(torch.int32).to(torch.int64) << 32) | rows[None]
    best = k.min(1).values
    gid = (rows >> 6) * 4 + ((rows >> 2) & 3)
    bg = gid[(best & 0xFFFFFFFF) - first_row]
    runner = k.masked_fill(gid[None] == bg[:, None], eng.KEY_EMPTY).min(1).values
    return torch.stack([best, runner])
if hi > lo:
    mine2 = top2(d2, lo)
else:
    mine2 = torch.full((2, qs.shape[0]), eng.KEY_EMPTY, dtype=torch.int64)
merged2 = eng.merge_shard_keys(mine2.clone(), td.group.WORLD)
want2 = top2(d2f, 0)
assert torch.equal(merged2, want2), (merged2 != want2).nonzero()[:6]
halves = [top2(d2f[:, a:b], a) for a, b in ((0, 512), (512, 1000))]
assert torch.equal(eng.merge_key_planes(*halves), want2)          # the same merge for shards replayed on one device
td.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_merge_world2_gloo(tmp_path):
    _run_gloo(_GLOO_WORKER.format(repo=REPO), tmp_path)


def test_isa_lint_main_loops():
    """tools/isa_lint.py on the compiled gfx950 ISA of the GEMM files (cross-compiled, no GPU): no vmcnt(0) between the
    LDS-DMA issue and the fragment reads of an MFMA loop, no scratch traffic in an MFMA block, no compiler-generated AGPR
    writes in the kernels that keep their accumulators in AGPRs by convention (gemm_wide.h)."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "isa_lint.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("main loops clean") == 5      # gemm, gemm_sk, l2min, conv, encoder_tail


def test_compat_shims_register_the_cuda_wheel_module_paths():
    code = ("import sys; sys.path.insert(0, %r); import cmdiad_amd.compat as c; c.install();"
            "from pointnet2_ops import pointnet2_utils; from knn_cuda import KNN;"
            "assert all(hasattr(pointnet2_utils, n) for n in ('furthest_point_sample', 'gather_operation', 'ball_query',"
            " 'grouping_operation', 'QueryAndGroup', 'GroupAll'));"
            "k = KNN(k=128, transpose_mode=True); assert k.k == 128; print('ok')") % REPO
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_install_dropin_redirects_reference_module_paths():
    code = ("import sys; sys.path.insert(0, %r); import cmdiad_amd; cmdiad_amd.install_dropin();"
            "from feature_extractors import multiple_features;"
            "from models.hallucination_network import HallucinationCrossModalityNetwork, HallucinationRGBFeatureToXYZInputMLP,"
            " HallucinationFeatureToInputConv, HallucinationCrossModalityConv; from models.hrnet import HRNet;"
            "import utils.lr_sched as l; from utils.utils import set_seeds, KNNGaussianBlur;"
            "from models.models import Model, PointTransformer, fps;"
            "assert multiple_features.DoubleRGBPointFeatures.__module__.startswith('cmdiad_amd');"
            "print('ok')") % REPO
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must START two ranks itself (round 1 parsed --gpus and
    ignored it).  --selftest-launch runs the launch path on the CPU: the parent spawns torch.distributed.run before anything
    touches a GPU, the two ranks join a gloo group, run the row-sharded merge and rank 0 prints one JSON line that the
    parent relays."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--selftest-launch"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["selftest_launch"] is True and rec["ranks"] == 2 and rec["merge_ok"] is True
    assert rec["merge"] == {"merge_ok": True} and rec["second"] == {"ranks_counted": 2} and rec["third"] == {"ranks_counted": 2}


@pytest.mark.parametrize("inject", ["second:1:raise", "second:0:raise", "second:1:hang", "merge:1:raise"])
def test_bench_line_survives_a_failing_leg(inject):
    """bench.py's secondary legs are fault-isolated (bench.LegRunner).  A gloo world of two runs the
    launcher's self-test with a failure injected into one leg on one rank -- an exception on the rank that prints, an exception on
    the OTHER rank (rank 0 then waits in a collective nobody completes), a rank that hangs: every time rank 0 prints ONE JSON line
    whose headline fields are intact, the failed leg carries an "error" (or, when the OTHER rank's failure was announced before
    rank 0 entered the leg, "skipped": both are correct records of a leg that did not run to its end), later collective legs are
    skipped or carry the time-out -- and the job ends with a NON-ZERO exit code (bench.EXIT_OUT_OF_STEP on the ranks; the launcher
    passes on torch.distributed.run's code): a hung collective or a failed rank must not look like success."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CMDIAD_BENCH_INJECT=inject, CMDIAD_BENCH_LEG_BUDGET="6")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--selftest-launch"],
                         capture_output=True, text=True, timeout=300, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.returncode != 0, "a run whose ranks fell out of step must not exit 0\n" + out.stderr[-3000:]
    rec = json.loads(lines[0])
    assert rec["metric"] == "selftest" and rec["value"] == 1.0 and rec["ranks"] == 2          # the headline survived
    leg, bad_rank = inject.split(":")[0], int(inject.split(":")[1])
    if bad_rank == 0:
        assert "error" in rec[leg], rec
    else:   # rank 0 either met the failure inside the collective (error / time-out) or heard of it before entering the leg (skipped)
        assert "error" in rec[leg] or "skipped" in rec[leg], rec
    if leg == "second":
        assert rec["merge"] == {"merge_ok": True}                                                # the leg before it is intact
    later = {"merge": ["second", "third"], "second": ["third"]}[leg]
    for name in later:
        assert name not in rec or "skipped" in rec[name] or "error" in rec[name], rec


def _write_pair_files(root, n=12, class_name="bagel"):
    """--save_frgb_xyz / --save_rgb_fxyz files of n samples through the drop-in's own writer (no extractor: sample i's tensors are
    filled with i, 100 + i, 200 + i, 300 + i, 400 + i) -- the same call tests/golden/make_golden.py:write_pair_files makes."""
    import types
    from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures
    me = types.SimpleNamespace(args=types.SimpleNamespace(save_frgb_xyz=True, save_rgb_fxyz=True, save_path_frgb_xyz=os.path.join(root, "frgb_xyz"),
                                                          save_path_rgb_fxyz=os.path.join(root, "rgb_fxyz")),
                               class_name=class_name, ins_id2=0, ins_id3=0, _engine=types.SimpleNamespace(xyz_patch=lambda ex, P: ex))
    for i in range(n):
        sample = (torch.full((1, 3, 224, 224), float(i)), torch.full((1, 3, 224, 224), 100.0 + i))
        DoubleRGBPointFeatures._save_pairs(me, [sample], torch.full((1, 784, 768), 300.0 + i), torch.full((1, 3136, 768), 200.0 + i),
                                           torch.full((1, 3136, 768), 400.0 + i), "train")
    return me


def test_pair_datasets_match_the_reference_over_files_the_dropin_wrote(tmp_path, golden):
    """VERDICT round 4, item 6 (input side): DoubleRGBPointFeatures' --save_frgb_xyz / --save_rgb_fxyz files
    (multiple_features.py:827-867) and the two pair datasets (dataset.py:268-362).  Golden G13 = the REFERENCE's dataset classes
    over files this package's writer produced (lengths, pairing under the string sort -- bagel10 before bagel2 --, element order,
    shapes); cmdiad_amd.dataset's classes must return the same from the same files.  Then PairRing: batches in the order the
    reference's DataLoader would draw under the same global seed."""
    from cmdiad_amd import dataset as ds
    me = _write_pair_files(str(tmp_path))
    assert me.ins_id2 == 12 and me.ins_id3 == 12
    names = sorted(os.listdir(tmp_path / "rgb_fxyz" / "train" / "fxyz"))
    assert names[:3] == ["bagel0_hfxyz.pt", "bagel0_lfxyz.pt", "bagel10_hfxyz.pt"] and len(names) == 24
    assert sorted(os.listdir(tmp_path / "frgb_xyz" / "train" / "xyz"))[1] == "bagel10_xyz.pt"
    assert os.path.isdir(tmp_path / "frgb_xyz" / "test" / "frgb") and os.path.isdir(tmp_path / "rgb_fxyz" / "test" / "rgb")
    lo = torch.load(tmp_path / "rgb_fxyz" / "train" / "fxyz" / "bagel3_lfxyz.pt")
    assert tuple(lo.shape) == (784, 768) and float(lo[0, 0]) == 303.0 and lo.dtype == torch.float32
    g = golden("g13_datasets.npz")
    ds.FeatureToInputPreTrainTensorDataset.device = "cpu"       # the class loads onto 'cuda' as the reference does
    try:
        for cls in ("FeatureToInputPreTrainTensorDataset", "InputToFeaturePreTrainTensorDataset"):
            for dt, sub in (("xyz_frgb", "frgb_xyz"), ("rgb_fxyz", "rgb_fxyz")):
                d = getattr(ds, cls)(str(tmp_path / sub / "train"), dt)
                want = g[f"{cls}.{dt}"]
                assert len(d) == len(want) == 12
                got = []
                for i in range(len(d)):
                    a, b = d[i]
                    got.append([float(a.flatten()[0]), a.dim(), a.shape[0], float(b.flatten()[0]), b.dim(), b.shape[0]])
                np.testing.assert_array_equal(np.array(got), want, err_msg=f"{cls}.{dt}")
        with pytest.raises(NotImplementedError):
            ds.InputToFeaturePreTrainTensorDataset(str(tmp_path / "rgb_fxyz" / "train"), "nope")
        # PairRing == DataLoader(dataset, shuffle=True, batch_size=5, drop_last=True) under the same seed, sample for sample
        d = ds.FeatureToInputPreTrainTensorDataset(str(tmp_path / "frgb_xyz" / "train"), "xyz_frgb")
        for shuffle, drop_last in ((True, True), (False, False)):
            torch.manual_seed(77)
            ref = [(a[:, 0, 0].tolist(), b[:, 0, 0, 0].tolist()) for a, b in
                   torch.utils.data.DataLoader(d, shuffle=shuffle, batch_size=5, drop_last=drop_last)]
            torch.manual_seed(77)
            ring = ds.PairRing(d, 5, shuffle=shuffle, drop_last=drop_last, device="cpu", readers=2)
            assert len(ring) == len(ref)
            got = [(a[:, 0, 0].tolist(), b[:, 0, 0, 0].tolist()) for a, b in ring]
            assert got == ref
            got2 = [(a[:, 0, 0].tolist(), b[:, 0, 0, 0].tolist()) for a, b in ring]      # second epoch: from the cache, new order
            assert len(got2) == len(got) and all(x - 400.0 == y - 100.0 for a, b in got2 for x, y in zip(a, b))   # pairs stay pairs
    finally:
        ds.FeatureToInputPreTrainTensorDataset.device = "cuda"


def test_save_raw_results_writes_the_reference_csv(tmp_path, monkeypatch):
    """--save_raw_results (features.py:316-318): calculate_metrics writes ./visualization/<note>/<class>_raw_results.csv with one
    row per test image -- score, label, image path -- before the metrics; the drop-in creates the directory."""
    import types
    from cmdiad_amd.feature_extractors.features import Features
    monkeypatch.chdir(tmp_path)
    rs = np.random.RandomState(0)
    n = 6
    labels = [np.array([i % 2]) for i in range(n)]
    gts = [np.zeros((16, 16), dtype=np.float32) for _ in range(n)]
    for i in range(1, n, 2):
        gts[i][4:8, 4:8] = 1.0
    preds = [rs.rand(16, 16).astype(np.float32) + g for g in gts]
    me = types.SimpleNamespace(args=types.SimpleNamespace(save_raw_results=True, experiment_note="note7"), class_name="bagel",
                               image_preds=[np.array([float(p.max())]) for p in preds], image_labels=labels,
                               pixel_preds=list(np.concatenate([p.ravel() for p in preds])), pixel_labels=list(np.concatenate([g.ravel() for g in gts])),
                               predictions=preds, gts=gts, img_name=[[f"/data/mvtec_3d/bagel/test/x/rgb/{i:03d}.png"] for i in range(n)])
    Features.calculate_metrics(me)
    rows = open(tmp_path / "visualization" / "note7" / "bagel_raw_results.csv").read().strip().splitlines()
    assert len(rows) == n and rows[1].split(",")[1] == "1" and rows[1].split(",")[2].endswith("001.png")
    assert abs(float(rows[3].split(",")[0]) - float(preds[3].max())) < 1e-5
    assert 0.5 < me.image_rocauc <= 1.0 and 0.5 < me.pixel_rocauc <= 1.0 and me.au_pro > 0


def test_empty_shard_cannot_win_the_min_reduce():
    """ADVICE (round 1): shard_range yields EMPTY shards when n <= 128 * (world - 1); such a rank contributes only the
    'no candidate' key, which must lose a signed MIN reduce against every real key (0xFFFF...F = -1 would have won)."""
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    assert ops.KEY_EMPTY == 0x7FFFFFFFFFFFFFFF and eng.KEY_EMPTY == ops.KEY_EMPTY
    ranges = [eng.shard_range(100, r, 8) for r in range(8)]
    assert ranges[0] == (0, 100) and all(lo == hi for lo, hi in ranges[1:])
    real = (torch.tensor([3.5e38, 0.0, 1.0]).view(torch.int32).to(torch.int64) << 32) | torch.tensor([99, 0, 0xFFFFFFFE])
    empty = torch.full((3,), ops.KEY_EMPTY, dtype=torch.int64)
    assert torch.equal(torch.minimum(real, empty), real)          # what all_reduce(MIN) computes
    assert int((empty & 0xFFFFFFFF)[0]) == 0xFFFFFFFF            # and it names no row of any library


def test_backbone_checkpoint_loading_is_strict(tmp_path, monkeypatch):
    """ADVICE (round 1, medium): a backbone must never silently stay at its random init.  No weights -> error unless
    CMDIAD_ALLOW_RANDOM_INIT=1; wrapped checkpoints ('model' / 'state_dict' / 'teacher', 'module.' / 'backbone.' prefixes)
    load; a checkpoint that lacks backbone tensors is an error, not an empty strict=False load."""
    from cmdiad_amd.models import models as mm
    monkeypatch.delenv("CMDIAD_VIT_CHECKPOINT", raising=False)
    monkeypatch.setenv("CMDIAD_ALLOW_RANDOM_INIT", "0")
    with pytest.raises(RuntimeError, match="no ViT-B/8 weights"):
        mm.Model("cpu")
    torch.manual_seed(1)
    sd = {k: v + 0.5 for k, v in mm.VisionTransformer().state_dict().items()}
    wrapped = tmp_path / "vit.pth"
    torch.save({"teacher": {"module.backbone." + k: v for k, v in sd.items()} | {"module.head.mlp.0.weight": torch.zeros(2)}}, wrapped)
    with pytest.raises(FileNotFoundError, match="pointmae"):          # the ViT loads, then Point-MAE has no checkpoint
        mm.Model("cpu", checkpoint_path=str(wrapped))
    monkeypatch.setenv("CMDIAD_ALLOW_RANDOM_INIT", "1")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = mm.Model("cpu", checkpoint_path=str(wrapped))
    for k, v in m.rgb_backbone.state_dict().items():
        assert torch.equal(v, sd[k]), k
    monkeypatch.setenv("CMDIAD_VIT_CHECKPOINT", str(wrapped))          # the environment variable is the other way in
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m2 = mm.Model("cpu")
    assert torch.equal(m2.rgb_backbone.pos_embed, sd["pos_embed"])
    monkeypatch.delenv("CMDIAD_VIT_CHECKPOINT")
    partial = tmp_path / "partial.pth"
    torch.save({k: v for k, v in sd.items() if not k.startswith("blocks.11.")}, partial)
    with pytest.raises(RuntimeError, match="lacks .* backbone tensors"):
        mm.Model("cpu", checkpoint_path=str(partial))
    # Point-MAE: the pretrain file's 'base_model' wrapper and MAE_encoder. prefix (models/models.py:285-295)
    torch.manual_seed(2)
    pm_sd = {k: v.clone() for k, v in mm.PointTransformer().state_dict().items()}
    pm_file = tmp_path / "pointmae_pretrain.pth"
    torch.save({"base_model": {"module.MAE_encoder." + k: v for k, v in pm_sd.items()} | {"module.MAE_decoder.x": torch.zeros(1)}}, pm_file)
    pt = mm.PointTransformer()
    pt.load_model_from_ckpt(str(pm_file))
    assert all(torch.equal(v, pm_sd[k]) for k, v in pt.state_dict().items())
    torch.save({"base_model": {"MAE_encoder.norm.weight": torch.ones(384)}}, pm_file)
    with pytest.raises(RuntimeError, match="lacks"):
        mm.PointTransformer().load_model_from_ckpt(str(pm_file))


def test_bench_traffic_is_tied_to_the_profiled_kernel_source(tmp_path, monkeypatch):
    """roofline.traffic comes from the newest committed PMC pass only while the distance GEMM's sources hash to what was profiled
    (profiles/rN_pmc_meta.json); any change to them makes it null instead of silently stale -- and the committed pass of this
    round IS of the committed sources."""
    import glob, importlib.util, json, re, shutil
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = bench.profiled_traffic()
    assert isinstance(t["traffic"], int) and 1e9 < t["traffic"] < 1e11 and "commit" in t["traffic_note"] and "traffic_regime" in t, t
    newest = max(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_meta.json")), key=lambda f: int(re.search(r"r(\d+)_pmc_meta", f).group(1)))
    tag = re.search(r"(r\d+)_pmc_meta", newest).group(1)
    assert f"profiles/{tag}_pmc.md" in t["traffic_note"]
    # a copy of the repo files with one byte appended to the kernel source
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    (root / "cmdiad_amd" / "csrc").mkdir(parents=True)
    meta = json.load(open(newest))
    for f in meta["sources"] + [f"profiles/{tag}_pmc.json", f"profiles/{tag}_pmc_meta.json"] + ([meta["standalone"]] if meta.get("standalone") else []):
        shutil.copy(os.path.join(REPO, f), root / f)
    with open(root / meta["sources"][0], "ab") as fh:
        fh.write(b"\n")
    assert bench.profiled_traffic(str(root))["traffic"] is None


def test_fast_score_samples_is_sklearns():
    """The drop-in's validation-free form of SGDOneClassSVM.score_samples (multiple_features._score_samples; features.py:352-358
    fits the fusers, multiple_features.py:1005-1006 scores with them) returns scikit-learn's own bits, for float32 maps as the
    method classes pass them, float64, non-contiguous input, and falls back to the method for anything it does not cover."""
    import numpy as np
    from sklearn import linear_model
    from cmdiad_amd.feature_extractors.multiple_features import _score_samples
    rs = np.random.RandomState(3)
    for n_fit in (64, 4096):
        f = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(n_fit, 2) * 3.0)
        for x in (rs.rand(50176, 2).astype(np.float32) * 5, rs.randn(1, 2).astype(np.float32), rs.rand(777, 2),
                  np.asfortranarray(rs.rand(1000, 2).astype(np.float32)), rs.rand(2000, 4).astype(np.float32)[:, ::2]):
            want = f.score_samples(x)
            got = _score_samples(f, x)
            assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)

    class Other:
        def score_samples(self, x):
            return "own method"
    assert _score_samples(Other(), np.zeros((2, 2), np.float32)) == "own method"
    f = linear_model.SGDOneClassSVM(random_state=0).fit(rs.rand(32, 2))
    bad = np.array([[np.nan, 1.0]], np.float32)
    try:
        f.score_samples(bad)
        raised = False
    except ValueError:
        raised = True
    try:
        _score_samples(f, bad)
        raised2 = False
    except ValueError:
        raised2 = True
    assert raised == raised2


def test_dropin_keeps_one_micro_batch_in_flight_and_drains_on_read(monkeypatch):
    """Host logic of the deferred micro-batches (multiple_features._MethodBase._defer / _flush / _complete), without a GPU: a full
    batch is scored (queued) at once but recorded only when the next one has been queued; reading a result attribute or a phase
    call records everything, in call order."""
    from cmdiad_amd.feature_extractors import multiple_features as mf
    monkeypatch.setenv("CMDIAD_PREDICT_BATCH", "2")
    scored = []

    class Fake(mf._MethodBase):
        def __init__(self):
            for name in ("s_lib", "s_map_lib", "image_preds", "img_name"):
                self.__dict__["_lz_" + name] = []

        def _score_batch(self, samples, test=False):
            scored.append(list(samples))
            return [(("s", x), ("map", x)) for x in samples]

        def _record(self, s, s_map, mask, label, rgb_path):
            self.image_preds.append(s)
            self.img_name.append(rgb_path)

    m = Fake()
    m.predict("a", None, 0, "a.png")
    assert scored == [] and m.__dict__["_lz_image_preds"] == []
    m.predict("b", None, 0, "b.png")
    assert scored == [["a", "b"]] and m.__dict__["_lz_image_preds"] == [] and "predict" in m.__dict__["_inflight"]
    m.predict("c", None, 0, "c.png")
    m.predict("d", None, 0, "d.png")
    # the second batch has been queued, the first one recorded meanwhile
    assert scored == [["a", "b"], ["c", "d"]] and m.__dict__["_lz_img_name"] == ["a.png", "b.png"]
    m.predict("e", None, 0, "e.png")
    assert m.img_name == ["a.png", "b.png", "c.png", "d.png", "e.png"]     # a read drains: batch in flight + the partial one
    assert scored[-1] == ["e"] and "predict" not in m.__dict__["_inflight"] and not m.__dict__["_pending"]["predict"]
    assert [s for s in m.image_preds] == [("s", x) for x in "abcde"]
    # the late-fusion phase has its own queue and flight slot
    for x in "xyz":
        m.add_sample_to_late_fusion_mem_bank(x)
    assert m.__dict__["_lz_s_lib"] == [] and "late" in m.__dict__["_inflight"]
    assert m.s_lib == [("s", "x"), ("s", "y"), ("s", "z")] and m.s_map_lib == [("map", "x"), ("map", "y"), ("map", "z")]


# ------------------------------------------------------------------------------------------------ class sharding (configs[4])
def test_lpt_assignment_covers_every_class_once_and_balances():
    """cmdiad_amd.evaluate.lpt_assign over the cost model of the ten MVTec 3D-AD classes: every class exactly once for any
    world size, deterministic, no rank above the LPT bound (4/3 - 1/(3m)) x optimum >= max(mean load, largest class)."""
    from cmdiad_amd import evaluate as ev
    costs = {c: ev.class_cost(ev.MVTEC3D_TRAIN[c], ev.MVTEC3D_TEST[c]) for c in ev.MVTEC3D_TRAIN}
    assert max(costs, key=costs.get) == "peach" and 5 < costs["bagel"] < 60          # seconds: coreset + SVM fit dominate
    for world in (1, 2, 3, 4, 8, 10, 16):
        a = ev.lpt_assign(costs, world)
        assert len(a) == world and sorted(sum(a, [])) == sorted(costs)
        assert a == ev.lpt_assign(dict(reversed(list(costs.items()))), world)          # independent of dictionary order
        loads = [sum(costs[c] for c in r) for r in a]
        lower = max(sum(costs.values()) / world, max(costs.values()))
        assert max(loads) <= (4 / 3 - 1 / (3 * world)) * lower * 1.25 + 1e-9, (world, loads)
    a8 = ev.lpt_assign(costs, 8)
    assert a8[0] == ["peach"] and sorted(len(r) for r in a8) == [1] * 6 + [2, 2]      # 10 classes on 8 ranks: two ranks take two
    assert ev.lpt_assign({"b": 1.0, "a": 1.0, "c": 1.0}, 2) == [["a", "c"], ["b"]]      # ties: by name, lowest rank first


def test_metrics_table_is_main_py_s_table():
    """main.py:27-37: per-class values rounded to 3 digits (cmdiad_runner.py:98-101), Mean = round(mean of those, 3)."""
    import pandas as pd
    from cmdiad_amd import evaluate as ev
    pc = {"bagel": dict(image_rocauc=0.91849, pixel_rocauc=0.99251, au_pro=0.9, au_pro_001=0.4),
          "cable_gland": dict(image_rocauc=0.7777, pixel_rocauc=0.5, au_pro=0.12345, au_pro_001=0.0)}
    t = ev.metrics_table(pc, "WithHallucination")
    df = pd.DataFrame(["WithHallucination"], columns=["Method"])                         # the reference's own construction
    for cls, v in pc.items():
        df[cls.title()] = df["Method"].map({"WithHallucination": round(v["image_rocauc"], 3)})
    df["Mean"] = round(df.iloc[:, 1:].mean(axis=1), 3)
    assert t["image_rocauc"] == df.iloc[0].to_dict()
    assert list(t["image_rocauc"]) == ["Method", "Bagel", "Cable_Gland", "Mean"]


_GLOO_EVAL_WORKER = r"""
import json, os, sys
sys.path.insert(0, {repo!r})
import torch.distributed as td
from cmdiad_amd import evaluate as ev
td.init_process_group("gloo")
rank = td.get_rank()
data = ev.synthetic_mvtec3d("all", scale=0.05, n_test=20)
calls = []
def runner(args, d, weights=None):      # stand-in for run_class: no GPU here; a class's result depends on the class alone
    calls.append(d.name)
    h = sum(map(ord, d.name))
    return dict(image_rocauc=(h % 97) / 97.0, pixel_rocauc=(h % 89) / 89.0, au_pro=(h % 83) / 83.0, au_pro_001=(h % 79) / 79.0,
                n_train=d.n_train, n_test=d.n_test, seconds=dict(predict=0.01 * d.n_test), library_rows=dict(xyz=d.n_train))
res = ev.evaluate_classes(ev.mtfi_args(), data, group=td.group.WORLD, runner=runner)
assert calls == res["assignment"][rank], (calls, res["assignment"])
print("RESULT " + json.dumps(dict(rank=rank, calls=calls, per_class=res["per_class"], table=res["table"], assignment=res["assignment"])))
td.destroy_process_group()
"""


def test_class_sharded_evaluate_world2_gloo(tmp_path):
    """cmdiad_amd.evaluate.evaluate_classes with a gloo group of two ranks (the N > 1 path of configs[4]; RCCL on the GPUs):
    the LPT assignment covers every class exactly once, each rank runs exactly its classes, and after the
    all_gather_object BOTH ranks hold the dictionary -- and the main.py table -- a single rank produces."""
    import json
    from cmdiad_amd import evaluate as ev
    outs = _run_gloo(_GLOO_EVAL_WORKER.format(repo=REPO), tmp_path, timeout=240)
    recs = [json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT ")][-1][7:]) for o in outs]
    recs.sort(key=lambda r: r["rank"])
    assert sorted(recs[0]["calls"] + recs[1]["calls"]) == sorted(ev.MVTEC3D_TRAIN) and not set(recs[0]["calls"]) & set(recs[1]["calls"])
    assert 4 <= len(recs[0]["calls"]) <= 6

    def runner(args, d, weights=None):
        h = sum(map(ord, d.name))
        return dict(image_rocauc=(h % 97) / 97.0, pixel_rocauc=(h % 89) / 89.0, au_pro=(h % 83) / 83.0, au_pro_001=(h % 79) / 79.0,
                    n_train=d.n_train, n_test=d.n_test, seconds=dict(predict=0.01 * d.n_test), library_rows=dict(xyz=d.n_train))
    single = ev.evaluate_classes(ev.mtfi_args(), ev.synthetic_mvtec3d("all", scale=0.05, n_test=20), runner=runner)
    for r in recs:
        assert r["table"] == single["table"] and list(r["per_class"]) == list(single["per_class"]) == list(ev.MVTEC3D_TRAIN)
        for cls, v in r["per_class"].items():
            want = dict(single["per_class"][cls], rank=v["rank"])
            assert v == want and cls in r["assignment"][v["rank"]]


# ------------------------------------------------------------------------------------------------ compact, then gather (configs[3])
_GLOO_COMPACT_WORKER = r"""
import os, sys, types
sys.path.insert(0, {repo!r})
import torch
import torch.distributed as td
from cmdiad_amd import engine as eng
td.init_process_group("gloo")
rank, world = td.get_rank(), td.get_world_size()


def pack(d2, idx):
    return (d2.contiguous().view(torch.int32).to(torch.int64) << 32) | idx


def d2_of(q, q_sq, b, b_sq):
    return ((q_sq[:, None] + b_sq[None, :]) - 2.0 * (q.float() @ b.float().T)).clamp_min(0.0)


class TorchSearch:      # stand-in for the HIP kernels behind engine.sharded_min_keys (same contracts, host tensors)
    searched = 0

    @staticmethod
    def plan(q16, q_sq, reuse=None):
        Q = q16.shape[0]
        _, inv, cnt = torch.unique(q16.float(), dim=0, return_inverse=True, return_counts=True)
        dup = inv == cnt.argmax()
        first = int(dup.nonzero()[0])
        keep = ~dup
        keep[first] = True
        rows = keep.nonzero().flatten()
        slot = (torch.cumsum(keep.int(), 0) - 1).int()
        slot[dup] = slot[first]
        p = types.SimpleNamespace(q16=torch.zeros_like(q16), q_sq=torch.zeros_like(q_sq), slot=slot,
                                  count=torch.tensor([len(rows)], dtype=torch.int32))
        p.q16[:len(rows)] = q16[rows]
        p.q_sq[:len(rows)] = q_sq[rows]
        return p

    @staticmethod
    def search(q16, q_sq, bank, keys):
        if bank.bf16.shape[0]:
            TorchSearch.searched += q16.shape[0]
            v, i = d2_of(q16, q_sq, bank.bf16, bank.sqnorm).min(1)
            keys.copy_(torch.minimum(keys, pack(v, i + bank.row_offset)))
        return keys

    @staticmethod
    def search_segments(q_all, s_all, counts, cap, bank, keys_all):
        for w, n in enumerate(counts.tolist()):       # host tensors here: reading the counts costs nothing
            n = min(n, cap)
            if n:
                TorchSearch.search(q_all[w * cap:w * cap + n], s_all[w * cap:w * cap + n], bank, keys_all[w * cap:w * cap + n])
        return keys_all

    @staticmethod
    def expand(kc, slot, out):
        out.copy_(kc[slot.long()])
        return out


g = torch.Generator().manual_seed(21)
Nb, D, Q = int(os.environ.get("T_NB", "1500")), 16, 700
lib = torch.randn(Nb, D, generator=g).half()                   # the same library on every rank; each keeps its row shard
lib_sq = lib.float().pow(2).sum(1)
lo, hi = eng.shard_range(Nb, rank, world)
if os.environ.get("T_EXPECT_EMPTY_LAST") == "1" and rank == world - 1:
    assert lo == hi == Nb                                      # 128-row aligned shards: the last rank holds NO library row
bank = types.SimpleNamespace(bf16=lib[lo:hi], sqnorm=lib_sq[lo:hi], row_offset=lo)
gq = torch.Generator().manual_seed(100 + rank)                 # every rank's own queries, with its own share of background rows
q = torch.randn(Q, D, generator=gq).half()
share = 0.35 + 0.2 * rank if world == 2 else 0.2 + 0.55 * rank / (world - 1)   # ragged: 20 % ... 75 % background rows at world 8
bg = torch.rand(Q, generator=gq) < share
q[bg] = torch.full((D,), -0.25).half()
q_sq = q.float().pow(2).sum(1)
stats = {{}}
keys, plan = eng.sharded_min_keys(q, q_sq, bank, td.group.WORLD, stats=stats, impl=TorchSearch)
# gather-then-search of EVERY row against the whole library: the single-device answer
v, i = d2_of(q, q_sq, lib, lib_sq).min(1)
want = pack(v, i)
assert torch.equal(keys, want), (keys != want).nonzero().flatten()[:8]
live = int(plan.count)
assert live == int((~bg).sum()) + 1 and stats["live_rows"][rank] == live
assert stats["gathered_rows_per_rank"] == min(Q, (max(stats["live_rows"]) + 255) // 256 * 256)
if world == 2:
    assert stats["gathered_rows_per_rank"] < Q and stats["gather_bytes_received"] < 0.8 * stats["gather_bytes_received_without_compaction"]
assert TorchSearch.searched == (sum(stats["live_rows"]) if hi > lo else 0)   # the live rows of all ranks, nothing else (nothing at all on an empty shard)

# steady state ("auto"): ONE host read for three steps; then a batch with more live rows than the sticky cap raises the flag
# on every rank, the repeated step (after regrow()) is exact again
ss = eng.ShardedSearch(bank, td.group.WORLD, impl=TorchSearch, cap_rows="auto", slack=0.0)
for step in range(3):
    k2 = ss.gather(q, q_sq).gemm().reduce()
    assert torch.equal(k2, want) and not ss.overflowed()
assert ss.host_reads == 1
q_more = q.clone()
if rank == 0:
    q_more[bg] = torch.randn(int(bg.sum()), D, generator=gq).half()     # rank 0's background rows become live rows
qm_sq = q_more.float().pow(2).sum(1)
cap_before = ss.cap
ss.gather(q_more, qm_sq).gemm().reduce()
grew = int(ss.counts_dev.max()) > cap_before
assert ss.overflowed() == grew and ss.host_reads == 1
if grew:
    ss.regrow()
    k3 = ss.gather(q_more, qm_sq).gemm().reduce()
    v3, i3 = d2_of(q_more, qm_sq, lib, lib_sq).min(1)
    assert torch.equal(k3, pack(v3, i3)) and not ss.overflowed() and ss.host_reads == 2 and ss.cap > cap_before
td.barrier()
td.destroy_process_group()
print("rank", rank, "ok", stats["live_rows"], stats["gathered_rows_per_rank"])
"""


def test_compact_then_gather_equals_gather_then_search_world2_gloo(tmp_path):
    """engine.sharded_min_keys on a gloo group of two ranks with a torch stand-in for the kernels: removing each rank's repeated
    background row BEFORE the all-gather (only the live rows + a per-rank count travel) returns, for every original row, the
    key that searching every row of every rank against the whole library returns -- ties to the lowest global row included."""
    outs = _run_gloo(_GLOO_COMPACT_WORKER.format(repo=REPO), tmp_path, timeout=240)


def test_compact_then_gather_world8_gloo_ragged_counts_and_an_empty_shard(tmp_path):
    """The same on a gloo world of EIGHT (BASELINE configs[3]'s rank count): ragged live counts (20 % ... 75 % background rows per
    rank), and a library so short (800 rows -> seven 128-row shards) that the last rank's shard is EMPTY -- its keys stay
    KEY_EMPTY and lose the MIN reduce."""
    outs = _run_gloo(_GLOO_COMPACT_WORKER.format(repo=REPO), tmp_path, world=8, timeout=600,
                     extra_env={"T_NB": "800", "T_EXPECT_EMPTY_LAST": "1"})
    assert all("ok" in o for o in outs), outs


_GLOO_EVAL_FAIL_WORKER = r"""
import sys
sys.path.insert(0, {repo!r})
import torch.distributed as td
from cmdiad_amd import evaluate as ev
td.init_process_group("gloo")
rank = td.get_rank()
data = ev.synthetic_mvtec3d(["bagel", "peach", "tire"], scale=0.05, n_test=20)
def runner(args, d, weights=None):
    if rank == 1:
        raise ValueError("cloud has 7 valid points")
    return dict(image_rocauc=1.0, pixel_rocauc=1.0, au_pro=1.0, au_pro_001=1.0, n_train=d.n_train, n_test=d.n_test, seconds={{}}, library_rows={{}})
try:
    ev.evaluate_classes(ev.mtfi_args(), data, group=td.group.WORLD, runner=runner)
    print("NO ERROR")
except RuntimeError as e:
    print("RAISED", e)
td.destroy_process_group()
"""


def test_class_sharded_evaluate_propagates_a_rank_failure_world2_gloo(tmp_path):
    """A rank whose class loop raises still reaches the gather, and EVERY rank then raises naming the rank and the cause -- no
    rank is left waiting in the collective until it times out."""
    outs = _run_gloo(_GLOO_EVAL_FAIL_WORKER.format(repo=REPO), tmp_path, timeout=120)
    for o in outs:
        assert "RAISED class-sharded evaluation failed on rank 1: ValueError: cloud has 7 valid points" in o, o


def test_layernorm_fold_weights_and_chain_flags():
    """Host logic of the LayerNorm fold (runtime.ln_fold / block_flags): folded weights reproduce LayerNorm -> Linear in float64 for any
    row mean; a block prepares the next block's first LayerNorm unless its output is read in between or it is the last."""
    import numpy as np
    import torch
    from cmdiad_amd import ops, runtime
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(50, 384, generator=g) * 2 + 5).double()
    W, b = torch.randn(96, 384, generator=g), torch.randn(96, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(384, generator=g), 0.2 * torch.randn(384, generator=g)
    Wf, bf = runtime.ln_fold(W, b, gamma, beta)
    assert torch.allclose(Wf.double().sum(1), torch.zeros(96, dtype=torch.float64), atol=1e-4)     # centred over k
    ref = torch.nn.functional.layer_norm(x, (384,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()
    rstd = 1.0 / torch.sqrt(x.var(dim=1, unbiased=False) + 1e-5)
    np.testing.assert_allclose((rstd[:, None] * (x @ Wf.double().T) + bf.double()).numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    Wn, bn = runtime.ln_fold(W, None, gamma, beta)                                                  # qkv without bias (Point-MAE)
    np.testing.assert_allclose(bn.numpy(), (W.double() @ beta.double()).float().numpy(), rtol=1e-6)
    R, P = ops.BLOCK_LN1_READY, ops.BLOCK_PREP_NEXT
    assert [runtime.block_flags(i, 4, True) for i in range(4)] == [P, R | P, R | P, R]
    assert [runtime.block_flags(i, 12, True, (3, 7, 11)) for i in range(12)] == [P, R | P, R | P, R, P, R | P, R | P, R, P, R | P, R | P, R]
    assert [runtime.block_flags(i, 4, False) for i in range(4)] == [0, 0, 0, 0]
    for v, want in (("0", (False, False)), ("1", (True, True)), ("pmae", (False, True)), ("vit", (True, False))):
        os.environ["CMDIAD_LN_FOLD"] = v
        try:
            assert (runtime.ln_fold_enabled("vit"), runtime.ln_fold_enabled("pmae")) == want
        finally:
            del os.environ["CMDIAD_LN_FOLD"]
    assert (runtime.ln_fold_enabled("vit"), runtime.ln_fold_enabled("pmae")) == (False, True)      # the default


def test_weight_gradient_row_slices():
    """conv_train._split_for: slices of the token dimension for few-tile weight-gradient products -- about two workgroups per CU,
    never more slices than pairs of 64-row steps, at most 64."""
    from cmdiad_amd import conv_train
    assert conv_train._split_for(107648, 768, 768) == 14          # 36 tiles x 14 = 504 workgroups
    assert conv_train._split_for(25088, 128, 128) == 64           # one tile: capped
    assert conv_train._split_for(25088, 512, 128) == 64           # four tiles: 256 workgroups at the cap
    assert conv_train._split_for(256, 128, 128) == 2 and conv_train._split_for(64, 64, 64) == 1
    assert conv_train._split_for(100352, 1920, 1920) == 2         # 225 tiles: already enough workgroups


# ------------------------------------------------------------------------------------------------ class loop: stages, cut-off, overlap
class _FakeMethod:
    """Protocol stand-in for a drop-in method class (cmdiad_runner.py:16-31): records (class, call, thread) and costs nothing."""
    log = []

    def __init__(self, args, shared_extractor=None):
        from sklearn.linear_model import SGDOneClassSVM
        self.deep_feature_extractor = shared_extractor or object()
        self.detect_fuser, self.seg_fuser = SGDOneClassSVM(), SGDOneClassSVM()    # scikit-learn estimators: a host-only fit
        self.image_preds, self.name = [], None

    def _note(self, what):
        import threading
        _FakeMethod.log.append((self.name, what, threading.current_thread() is threading.main_thread()))

    def add_sample_to_mem_bank(self, sample, class_name=None):
        self.name = class_name
        self._note("bank")

    def run_coreset(self):
        self._note("coreset")

    def add_sample_to_late_fusion_mem_bank(self, sample):
        self._note("late")

    def run_late_fusion(self):
        import time
        self._note("fit_begin")
        time.sleep(0.3)
        self._note("fit_end")

    def predict(self, sample, mask, label, rgb_path):
        self.image_preds.append(0.0)
        self._note("predict")

    def calculate_metrics(self):
        self._note("metrics")
        self.image_rocauc = self.pixel_rocauc = self.au_pro = self.au_pro_001 = 0.5


def _fake_eval(monkeypatch):
    from cmdiad_amd import evaluate as ev
    _FakeMethod.log = []
    monkeypatch.setattr(ev, "method_class", lambda a: ("WithHallucination", _FakeMethod))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return ev


def test_max_sample_cuts_the_train_loops_only(monkeypatch):
    """cmdiad_runner.py:43-66 stop the two TRAIN loops after max_sample + 1 samples; cmdiad_runner.py:80-85 predicts EVERY test
    sample (round 3 truncated the test loop too: metrics on a truncated test set)."""
    ev = _fake_eval(monkeypatch)
    data = ev.synthetic_mvtec3d(["carrot"], n_train=5, n_test=4)["carrot"]
    res = ev.run_class(ev.mtfi_args(max_sample=1), data)
    calls = [c for _, c, _ in _FakeMethod.log]
    assert calls.count("bank") == 2 and calls.count("late") == 2 and calls.count("predict") == 4
    assert res["n_train"] == 2 and res["n_test"] == 4
    assert res["phases"] == ["memory_bank", "coreset", "late_fusion_bank", "late_fusion_fit", "predict", "metrics"]


def test_overlapped_class_schedule_keeps_every_class_in_order(monkeypatch):
    """evaluate.run_classes_overlapped: the host one-class-SVM fits of class k run on a worker thread beside the device stage of
    class k + 1; inside every class the reference's phase order (cmdiad_runner.py:33-107) is untouched, predict(k) waits for
    fit(k), and the results equal the in-line loop's."""
    ev = _fake_eval(monkeypatch)
    data = ev.synthetic_mvtec3d(["bagel", "peach", "tire"], n_train=2, n_test=3)
    names = ["peach", "bagel", "tire"]
    out = ev.run_classes_overlapped(ev.mtfi_args(), data, names)
    log = list(_FakeMethod.log)
    want = ["bank", "bank", "coreset", "late", "late", "fit_begin", "fit_end", "predict", "predict", "predict", "metrics"]
    for cls in names:
        assert [c for n, c, _ in log if n == cls] == want, cls                     # per class: the reference's order
        assert out[cls]["phases"] == ["memory_bank", "coreset", "late_fusion_bank", "late_fusion_fit", "predict", "metrics"]
    assert all(main for _, c, main in log if not c.startswith("fit"))              # device stages + predict: the main thread
    assert not any(main for _, c, main in log if c.startswith("fit"))              # the SVM fits: the worker
    pos = {(n, c): i for i, (n, c, _) in enumerate(log)}                           # (last occurrence)
    first = {}
    for i, (n, c, _) in enumerate(log):
        first.setdefault((n, c), i)
    # the fit of class k overlaps the device stage of class k + 1, and predict(k) follows both
    assert first[("peach", "fit_begin")] < first[("bagel", "bank")] < pos[("peach", "fit_end")]
    assert pos[("bagel", "late")] < first[("peach", "predict")] and pos[("peach", "fit_end")] < first[("peach", "predict")]
    assert first[("bagel", "fit_begin")] < first[("tire", "bank")] and pos[("tire", "fit_end")] < first[("tire", "predict")]
    _FakeMethod.log = []
    inline = {c: ev.run_class(ev.mtfi_args(), data[c]) for c in names}
    for c in names:
        a, b = dict(out[c]), dict(inline[c])
        for d in (a, b):
            d.pop("seconds"), d.pop("_extractor", None)
        assert a == b


_GLOO_DRIVE_WORKER = r"""
import sys
sys.path.insert(0, {repo!r})
import torch
import torch.distributed as td
from cmdiad_amd import engine as eng
td.init_process_group("gloo")
rank, world = td.get_rank(), td.get_world_size()


def steps():
    # the contract of engine._sharded_score_steps: "sum" -> element-wise sum over the ranks (one owner, zeros elsewhere),
    # "gather" -> [W, *shape] of every rank's tensor
    owned = torch.zeros(6)
    owned[rank::world] = torch.arange(6, dtype=torch.float32)[rank::world] + 1.0      # every element has exactly one owner
    total = yield ("sum", owned)
    assert torch.equal(total, torch.arange(6, dtype=torch.float32) + 1.0)
    mine = torch.tensor([[10 * rank + 3, 10 * rank + 1, 10 * rank + 2]], dtype=torch.int64)   # per-rank top-3 keys of one probe
    everyone = yield ("gather", mine)
    assert everyone.shape == (world, 1, 3)
    merged = everyone.permute(1, 0, 2).reshape(1, -1).sort(1).values[:, :3]
    assert merged.tolist() == [[1, 2, 3]]
    return "done"


assert eng._drive_collectives(steps(), td.group.WORLD) == "done"
td.barrier()
td.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_reweight_exchange_protocol_world2_gloo(tmp_path):
    """engine._drive_collectives (what runs engine._sharded_score_steps over RCCL: SURVEY 8(e)'s re-weight step on a library whose
    fp32 rows are sharded) on a gloo world of two: "sum" is an all_reduce(SUM), "gather" an all-gather into [W, ...]; the generator's
    return value comes back.  The arithmetic of the steps themselves is checked on the GPU (tests/test_gpu_fakeworld.py)."""
    outs = _run_gloo(_GLOO_DRIVE_WORKER.format(repo=REPO), tmp_path)
    assert all("ok" in o for o in outs), outs


_GLOO_CORESET_WORKER = r"""
import sys
sys.path.insert(0, {repo!r})
import torch
import torch.distributed as td
from cmdiad_amd import coreset
td.init_process_group("gloo")
rank, world = td.get_rank(), td.get_world_size()


class TorchRounds:      # stand-in for the HIP kernels behind coreset.greedy_coreset_sharded (same contracts, host tensors):
    # features.py:372-425 with coreset_dtype 'FP16' -- difference rounded to half, norm accumulated in float, result rounded to half
    def __init__(self, z):
        self.n, self.d = z.shape
        self.z = z.half()
        self.min_d = torch.linalg.norm(z - z[0:1], dim=1).half()

    def round(self, lo, hi, pivot_key, out_key):
        last = 0 if pivot_key is None else int(0xFFFFFFFF - (int(pivot_key[0]) & 0xFFFFFFFF))
        if hi == lo:
            return
        dist = torch.linalg.norm((self.z[lo:hi] - self.z[last:last + 1]).float(), dim=1).half()
        self.min_d[lo:hi] = torch.minimum(dist, self.min_d[lo:hi])
        m = self.min_d[lo:hi].float()
        i = int(torch.argmax(m))                                       # first occurrence = lowest row
        key = (int(m[i].view(torch.int32)) << 32) | (0xFFFFFFFF - (lo + i))
        out_key[0] = max(int(out_key[0]), key)

    def decode(self, keys, n_select):
        return torch.tensor([0] + [0xFFFFFFFF - (int(k) & 0xFFFFFFFF) for k in keys[:n_select - 1]])


g = torch.Generator().manual_seed(13)
z = torch.randn(1003, 40, generator=g)
z[900] = z[5]                                                          # a duplicate in another shard
got = coreset.greedy_coreset_sharded(z, 60, td.group.WORLD, impl=TorchRounds)
# the reference's loop in one process (features.py:372-425)
zh, min_d, last, want = z.half(), torch.linalg.norm(z - z[0:1], dim=1).half(), 0, [0]
for _ in range(59):
    min_d = torch.minimum(torch.linalg.norm((zh - zh[last:last + 1]).float(), dim=1).half(), min_d)
    last = int(torch.argmax(min_d.float()))
    min_d[last] = 0
    want.append(last)
assert got.tolist() == want, (got.tolist()[:10], want[:10])
lo, hi = coreset.shard_rows(1003, rank, world)
assert lo % 4 == 0 and (hi % 4 == 0 or hi == 1003)
td.barrier()
td.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_coreset_exchange_gloo(tmp_path, world):
    """coreset.greedy_coreset_sharded on a gloo group (torch stand-in for the scan kernel): per round every rank scans its own
    4-row aligned range and ONE all_reduce(MAX) of the packed (running minimum, ~row) key picks the pivot -- the selection equals
    the reference's single loop (features.py:372-425), ties between duplicate rows in different shards included."""
    outs = _run_gloo(_GLOO_CORESET_WORKER.format(repo=REPO), tmp_path, world=world)
    assert all("ok" in o for o in outs), outs


def test_dropin_serves_every_name_the_reference_scripts_take_from_redirected_modules(golden):
    """The reference's dataset.py / main.py / cmdiad_runner.py / hallucination_network_pretrain.py are NOT replaced, and they import
    from modules that install_dropin() redirects to this package (`from utils.mvtec3d_util import *`, dataset.py:9): every name
    they take (tests/golden/make_golden.py G14, from the reference's import statements) must exist here.  Round 5 found three
    missing: the redirected utils.mvtec3d_util had only organized_pc_to_unorganized_pc, so the reference's own dataset would have
    raised NameError on its first sample."""
    import importlib
    import cmdiad_amd
    g = golden("g14_surface.npz")
    names = [str(n) for n in g["names"]]
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(importlib.import_module(cmdiad_amd._DROPIN[n.split(":")[0]]), n.split(":")[1])]
    assert not missing, missing


def test_resize_organized_pc_equals_the_reference(golden):
    """utils/mvtec3d_util.py:14-26 on seeded scans with invalid (all-zero) points: nearest-neighbour resize to smaller, larger and
    equal grids, both output forms, and the depth channel -- bit for bit the reference's outputs (G14), and torch's mode='nearest'
    on MVTec-sized scans (800 x 800 -> 224 x 224)."""
    from cmdiad_amd.utils import mvtec3d_util as mv
    g = golden("g14_surface.npz")
    for i in range(4):
        scan, (h, w) = g[f"scan_{i}"], g[f"size_{i}"]
        t = mv.resize_organized_pc(scan, target_height=int(h), target_width=int(w))
        assert t.is_contiguous() and t.dtype == torch.float32 and np.array_equal(t.numpy(), g[f"tensor_{i}"])
        a = mv.resize_organized_pc(scan, target_height=int(h), target_width=int(w), tensor_out=False)
        assert isinstance(a, np.ndarray) and np.array_equal(a, g[f"array_{i}"])
        assert np.array_equal(mv.organized_pc_to_depth_map(scan), g[f"depth_{i}"])
    rs = np.random.RandomState(3)
    for H, W in ((800, 800), (400, 400), (777, 333)):
        scan = rs.randn(H, W, 3).astype(np.float32)
        ref = torch.nn.functional.interpolate(torch.tensor(scan).permute(2, 0, 1).unsqueeze(0).contiguous(), size=(224, 224), mode="nearest")[0]
        assert torch.equal(mv.resize_organized_pc(scan), ref)
    flat = mv.organized_pc_to_unorganized_pc(scan)
    assert flat.shape == (H * W, 3) and np.array_equal(flat[W + 2], scan[1, 2])
    with pytest.raises((ImportError, FileNotFoundError, OSError, ValueError)):
        mv.read_tiff_organized_pc("/nonexistent/file.tiff")


@pytest.mark.parametrize("order", ["path_first", "install_first", "no_reference"])
def test_install_dropin_is_order_independent(order):
    """install_dropin() before or after the reference's tree enters sys.path, or with no reference at all: the reference's own
    unreplaced scripts import (dataset.py star-imports a redirected module; hallucination_network_pretrain.py needs the reference's
    OWN utils.misc beside the redirected utils.lr_sched), redirected names resolve here.  Round 5: installed first, the old
    install_dropin() registered an empty `utils` package and the reference's utils.misc was gone."""
    ref = os.environ.get("CMDIAD_REFERENCE", "/root/reference")
    if order != "no_reference" and not os.path.isdir(ref):
        pytest.skip("the reference's tree is not on this machine")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", CMDIAD_ALLOW_RANDOM_INIT="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dropin_order_check.py"), order, ref], capture_output=True, text=True,
                         timeout=300, env=env, cwd="/tmp")
    assert out.returncode == 0 and f"{order} ok" in out.stdout, out.stderr[-2500:]


# names of the redirected modules that this package deliberately does not carry (each with its reason); everything else of the
# reference's 143 top-level functions, classes and methods must exist with the same positional argument names
_SURFACE_OMISSIONS = {
    "models.models|PointTransformer.load_model_from_pb_ckpt": "Point-BERT backbone (encoder_dims 256): out of scope, PointTransformer raises for it",
    "models.hrnet|BasicBlock": "building block of the reference's HRNet composition; HRNet here keeps the state_dict keys, not the classes",
    "models.hrnet|StageModule": "as BasicBlock",
    "utils.au_pro_util|GroundTruthComponent": "internal of calculate_au_pro (vectorised here)",
    "utils.au_pro_util|collect_anomaly_scores": "internal of calculate_au_pro",
    "utils.au_pro_util|compute_pro": "internal of calculate_au_pro",
}


def test_dropin_modules_carry_the_reference_surface(golden):
    """Every top-level function, class and method of the ten redirected reference modules (G14: names and positional argument
    names, from the reference's source) exists in the module that replaces it with the same leading arguments -- except the six
    listed omissions.  Round 5 found Features.interpolate_points, DepthFeatures and utils.utils.Interpolate missing."""
    import importlib
    import inspect
    import cmdiad_amd
    sigs = [str(x) for x in golden("g14_surface.npz")["signatures"]]
    assert len(sigs) >= 140
    problems = []
    for line in sigs:
        mod, qual, args = line.split("|")
        if any(f"{mod}|{qual}" == k or f"{mod}|{qual}".startswith(k + ".") for k in _SURFACE_OMISSIONS):
            continue
        obj = importlib.import_module(cmdiad_amd._DROPIN[mod])
        try:
            for part in qual.split("."):
                obj = getattr(obj, part)
        except AttributeError:
            problems.append(f"missing: {mod}.{qual}")
            continue
        if not args or args.endswith("*"):
            continue
        want = args.split(",")
        try:
            params = list(inspect.signature(obj).parameters.values())
        except (TypeError, ValueError):
            continue
        if any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in params):
            continue          # (*a, **k) pass-throughs accept the reference's arguments whatever their names
        have = [p.name for p in params]
        if have[:len(want)] != want:
            problems.append(f"arguments of {mod}.{qual}: reference {want}, here {have}")
    assert not problems, problems
