#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE (evenrose/CMDIAD) in the build container.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box: only the .npz fixtures written next to this
script travel).  Third-party packages the reference imports but this image lacks are
replaced by inert stubs; the two CUDA-only ops the reference reaches through them
(furthest_point_sample / KNN) are served by this repo's CPU oracle, so goldens that pass
through them pin everything *after* those ops, not the ops themselves (oracle/README.md).

Fixtures hold data only: seeded inputs (when small) and the reference's outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CMDIAD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from oracle import kernels as ok  # noqa: E402
from oracle import nets as onets  # noqa: E402
from cmdiad_amd.synth import synth_cloud  # noqa: E402


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(torch.nn.Module):  # eval-mode semantics: identity
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    timm = mod("timm", create_model=lambda **kw: (_ for _ in ()).throw(RuntimeError("timm stub")))
    mod("timm.models", layers=None)
    mod("timm.models.layers", DropPath=DropPath)
    timm.models = sys.modules["timm.models"]
    timm.models.layers = sys.modules["timm.models.layers"]

    class KNN:
        def __init__(self, k, transpose_mode=True):
            self.k = k

        def __call__(self, ref, query):  # ref [B,N,3], query [B,G,3]
            idx, _ = ok.knn_group(ref.numpy(), query.numpy(), self.k)
            return None, torch.from_numpy(idx)

    mod("knn_cuda", KNN=KNN)

    def furthest_point_sample(xyz, n):
        idx, _ = ok.fps(xyz.numpy(), n)
        return torch.from_numpy(idx)

    def gather_operation(feat, idx):  # feat [B,C,N], idx [B,G] -> [B,C,G]
        return torch.gather(feat, 2, idx.long().unsqueeze(1).expand(-1, feat.shape[1], -1))

    p2 = mod("pointnet2_ops")
    p2.pointnet2_utils = mod("pointnet2_ops.pointnet2_utils", furthest_point_sample=furthest_point_sample,
                             gather_operation=gather_operation)
    mod("cupy", asarray=lambda x: x)
    mod("cupyx")
    mod("cupyx.scipy")
    mod("cupyx.scipy.spatial", distance=None)
    mod("tifffile")

    # torchvision: only ToPILImage / ToTensor are reached (utils/utils.py:75-76)
    from PIL import Image

    class ToPILImage:
        def __call__(self, pic):  # float tensor [1,H,W] -> 'L' image via mul(255).byte()
            return Image.fromarray(pic.mul(255).byte().squeeze(0).numpy(), mode="L")

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).float().div(255).unsqueeze(0)

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", ToPILImage=ToPILImage, ToTensor=ToTensor,
                        v2=mod("torchvision.transforms.v2"))


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def golden_heads():
    """G10: the conv / feature-to-input / HRNet distillation heads (models/hallucination_network.py:72-220,
    models/hrnet.py), eval mode, on the synthetic weights of oracle.heads.synth_head_state_dict: outputs of
    hallucination_generation (sub-sampled), the loss each forward() returns, and checksums of a seeded default init."""
    from oracle import heads as oh
    from models import hallucination_network as rhn
    from models.hrnet import HRNet as RefHRNet
    g = torch.Generator().manual_seed(101)
    xyz_tok, rgb_tok = torch.randn(1, 3136, 768, generator=g), torch.randn(1, 3136, 768, generator=g)
    img = torch.randn(1, 3, 224, 224, generator=g)
    out = dict(tok_seed=101)
    build = {"conv_ftof": lambda: rhn.HallucinationCrossModalityConv(None, 768, 768),
             "ftoi_mlp": lambda: rhn.HallucinationRGBFeatureToXYZInputMLP(_ns(estimate_depth=False), 768),
             "ftoi_conv": lambda: rhn.HallucinationFeatureToInputConv(None, 768),
             "hrnet": lambda: RefHRNet(512, 768, 0.1)}
    with torch.no_grad():
        for kind, make in build.items():
            torch.manual_seed(777)
            m = make()
            for k, v in m.state_dict().items():  # seeded default init: construction order == RNG consumption order
                out[f"init/{kind}/{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
            sd = oh.synth_head_state_dict(kind, 41)
            print(kind, m.load_state_dict(sd))
            m.eval()
            if kind == "conv_ftof":
                xh, rh = m.hallucination_generation(xyz_tok, rgb_tok, "train")
                assert torch.allclose(xh, oh.conv_ftof(sd, rgb_tok, "rgb"), atol=1e-4) and torch.allclose(rh, oh.conv_ftof(sd, xyz_tok, "xyz"), atol=1e-4)
                out["conv_ftof/xyz_h"], out["conv_ftof/rgb_h"] = xh[0, ::7, ::8].numpy(), rh[0, ::7, ::8].numpy()
                out["conv_ftof/loss"] = np.array([float(v) for v in m(xyz_tok, rgb_tok, False, "l2")])
                out["conv_ftof/loss_sigmoid"] = np.array([float(v) for v in m(xyz_tok, rgb_tok, True, "l2")])
            elif kind == "ftoi_mlp":
                y = m.hallucination_generation(rgb_tok)
                assert torch.allclose(y, oh.ftoi_mlp(sd, rgb_tok), atol=1e-5)
                out["ftoi_mlp/y"], out["ftoi_mlp/loss"] = y[0, :, ::4, ::4].numpy(), float(m(rgb_tok, img))
            elif kind == "ftoi_conv":
                y = m.hallucination_generation(xyz_tok)
                assert torch.allclose(y, oh.ftoi_conv(sd, xyz_tok), atol=1e-4)
                out["ftoi_conv/y"], out["ftoi_conv/loss"] = y[0, :, ::4, ::4].numpy(), float(m(xyz_tok, img))
            else:
                y = m.hallucination_generation(img)
                assert torch.allclose(y, oh.hrnet(sd, img), atol=1e-4)
                out["hrnet/y"], out["hrnet/loss"] = y[0, ::8, ::2, ::2].numpy(), float(m(img, xyz_tok))
    np.savez_compressed(os.path.join(HERE, "g10_heads.npz"), **out)
    print("g10_heads.npz", os.path.getsize(os.path.join(HERE, "g10_heads.npz")) // 1024, "KB")


def _train_inputs(kind):
    """Seeded batch of two samples in the shapes the pretraining loop feeds each head (hallucination_network_pretrain.py:106-140)."""
    g = torch.Generator().manual_seed(202)
    a, b = torch.randn(2, 3136, 768, generator=g), torch.randn(2, 3136, 768, generator=g)
    img = torch.randn(2, 3, 224, 224, generator=g)
    return {"conv_ftof": (a, b), "ftoi_mlp": (a, img), "ftoi_conv": (a, img), "hrnet": (img, b)}[kind]


def golden_heads_train():
    """G12: the reference's own conv / feature-to-input / HRNet heads in TRAINING mode (batch-statistics BatchNorm, autograd) from
    the synthetic weights of oracle.heads.synth_head_state_dict, three torch.optim.Adam steps (hallucination_network_pretrain.py:261,
    lr 1e-4) on one seeded batch of two: the loss before every step and checksums of every tensor of the state_dict afterwards."""
    from oracle import heads as oh
    from models import hallucination_network as rhn
    from models.hrnet import HRNet as RefHRNet
    build = {"conv_ftof": lambda: rhn.HallucinationCrossModalityConv(None, 768, 768),
             "ftoi_mlp": lambda: rhn.HallucinationRGBFeatureToXYZInputMLP(_ns(estimate_depth=False), 768),
             "ftoi_conv": lambda: rhn.HallucinationFeatureToInputConv(None, 768),
             "hrnet": lambda: RefHRNet(512, 768, 0.1)}
    out = dict(input_seed=202, weight_seed=41, lr=1e-4, steps=3)
    torch.set_num_threads(8)
    for kind, make in build.items():
        m = make()
        m.load_state_dict(oh.synth_head_state_dict(kind, 41))
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=float(out["lr"]))
        x = _train_inputs(kind)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            if kind == "conv_ftof":
                lx, lr_ = m(x[0], x[1], False, "l2")
                loss = lx + lr_
                losses.append([float(lx), float(lr_)])
            else:
                loss = m(x[0], x[1])
                losses.append([float(loss)])
            loss.backward()
            opt.step()
            print(kind, losses[-1], flush=True)
        out[f"{kind}/loss"] = np.array(losses)
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                out[f"{kind}/after/{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    np.savez_compressed(os.path.join(HERE, "g12_heads_train.npz"), **out)
    print("g12_heads_train.npz", os.path.getsize(os.path.join(HERE, "g12_heads_train.npz")) // 1024, "KB")


class _CpuRedirect:
    """The reference hard-codes CUDA placement (features.py:96,103,106,397-399; multiple_features.py:327,334,344,350):
    inside this context `.cuda()` and `.to("cuda")` are no-ops so that its own arithmetic runs on the CPU (generator-only
    patch; nothing of it is shipped)."""

    def __enter__(self):
        self.to, self.tcuda, self.mcuda = torch.Tensor.to, torch.Tensor.cuda, torch.nn.Module.cuda
        orig_to = self.to
        torch.Tensor.to = lambda t, *a, **k: orig_to(t, *[("cpu" if (isinstance(x, str) and x == "cuda") else x) for x in a], **k)
        torch.Tensor.cuda = lambda t, *a, **k: t
        torch.nn.Module.cuda = lambda m, *a, **k: m
        return self

    def __exit__(self, *a):
        torch.Tensor.to, torch.Tensor.cuda, torch.nn.Module.cuda = self.to, self.tcuda, self.mcuda
        return False


def method_args(**kw):
    a = dict(rgb_backbone_name="vit_base_patch8_224_dino", xyz_backbone_name="Point_MAE", group_size=128, num_group=1024,
             rgb_size=224, xyz_size=224, gt_size=224, f_coreset=0.25, coreset_eps=0.9, coreset_dtype="FP16",
             random_state=0, dist_method_s="l2", dist_method_coreset="l2", main_modality="", use_hn=False,
             use_hn_conv=False, use_hn_from_rgb_mlp=False, use_hn_from_rgb_conv=False, use_hrnet=False, use_uff=False,
             use_depth=False, fusion_module_path="", ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0,
             xyz_smap_lambda=1.0, rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0,
             save_feature_for_fusion=False, save_frgb_xyz=False, save_rgb_fxyz=False, save_seg_results=False,
             save_raw_results=False, save_path="", save_path_frgb_xyz="", save_path_rgb_fxyz="", experiment_note="", c_hrnet=0)
    a.update(kw)
    return _ns(**a)


def golden_methods():
    """G11: the five-call protocol (cmdiad_runner.py:44-92) of the reference's OWN RGBFeatures (multiple_features.py:28-121),
    PointFeatures (:207-309) and RGBorXYZWithOneHallucination (:312-573; --use_hn, main modality xyz and rgb) over this
    repo's CPU backbone restatements (timm / pointnet2_ops / knn_cuda are absent) -- as G6 does for DoubleRGBPointFeatures.
    Point-MAE runs with the 'sharpened' synthetic weights (oracle.nets.sharpen_pointmae) so the xyz nearest-neighbour
    distances are genuine numbers, not fp32 cancellation noise.  Pins oracle/pipeline.py's CpuSingleModality /
    CpuOneHallucination."""
    from feature_extractors import features as rfeat
    from feature_extractors import multiple_features as rmf
    from cmdiad_amd.synth import synth_rgb
    sd_vit = onets.synth_state_dict("vit", 31)
    sd_pm = onets.sharpen_pointmae(onets.synth_state_dict("pointmae", 21))
    sd_h = onets.synth_state_dict("halluc", 51)
    cache = {}

    class FakeModel(torch.nn.Module):
        def __init__(self, device, rgb_backbone_name, xyz_backbone_name, group_size, num_group):
            super().__init__()
            self.G, self.M = num_group, group_size

        def forward(self, rgb, xyz, out_type="rgb+xyz"):
            key = (float(rgb.double().sum()), float(xyz.double().sum()), tuple(xyz.shape))
            if key not in cache:
                with torch.no_grad():
                    fmap = onets.vit_forward(sd_vit, rgb)
                    pts = np.ascontiguousarray(xyz[0].T.numpy())[None]
                    cidx, cen = ok.fps(pts, self.G)
                    idx, nb = ok.knn_group(pts, cen, self.M)
                    center = torch.from_numpy(cen)
                    tok = onets.pointmae_encoder(sd_pm, torch.from_numpy(nb))
                    feats = onets.pointmae_transformer(sd_pm, tok, center)
                cache[key] = (fmap, feats, center, torch.from_numpy(idx), torch.from_numpy(cidx))
            return cache[key]

    rfeat.Model = FakeModel
    TRAIN, TEST = (301, 302, 303), (311, 312)
    FRAC, TEX = 0.45, 0.004

    def sample(sd, anomalous=False):
        pc = synth_cloud(sd, FRAC, texture=TEX)
        rgb = synth_rgb(sd)
        if anomalous:
            pc[0, 2, 100:120, 100:120] -= 0.005 * (pc[0, 2, 100:120, 100:120] != 0)
            rgb[0, :, 100:120, 100:120] += 2.0
        return (rgb, pc, pc.clone())

    out = dict(train_seeds=np.array(TRAIN), test_seeds=np.array(TEST), test_anomalous=np.array([0, 1]), frac=FRAC, texture=TEX,
               f_coreset=0.25, random_state=0, pm_conv_gain=400.0, pm_qk_gain=36.0)
    runs = {"rgb": (rmf.RGBFeatures, {}), "xyz": (rmf.PointFeatures, {}),
            "mtfi_xyz": (rmf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality="xyz")),
            "mtfi_rgb": (rmf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality="rgb"))}
    for tag, (cls, kw) in runs.items():
        with _CpuRedirect():
            m = cls(method_args(**kw))
            if kw.get("use_hn"):
                print(tag, m.fusion.load_state_dict(sd_h))
            for sd in TRAIN:
                m.add_sample_to_mem_bank(sample(sd), class_name="synthetic")
            picked = []
            inner = m.get_coreset_idx_randomp
            m.get_coreset_idx_randomp = lambda *a, **k: (picked.append(inner(*a, **k)), picked[-1])[1]
            m.run_coreset()
            for sd in TRAIN:
                m.add_sample_to_late_fusion_mem_bank(sample(sd))
            s_lib, s_map_lib = torch.cat(m.s_lib, 0).clone(), torch.cat(m.s_map_lib, 0).clone()
            m.run_late_fusion()
            for sd, an in zip(TEST, (False, True)):
                m.predict(sample(sd, an), torch.zeros(1, 224, 224), 0, ["x.png"])
        g = {"s_lib": s_lib.numpy(), "s_map_lib_sub": s_map_lib[::53].numpy(),
             "detect_coef": m.detect_fuser.coef_, "detect_offset": m.detect_fuser.offset_,
             "seg_coef": m.seg_fuser.coef_, "seg_offset": m.seg_fuser.offset_,
             "image_preds": np.array(m.image_preds).reshape(-1), "pred_maps_sub": np.array(m.predictions)[:, ::4, ::4]}
        for k, idx in enumerate(picked):
            g[f"coreset_idx{k}"] = idx.numpy().astype(np.int32)
        if tag == "rgb":
            g.update(mean=float(m.rgb_mean), std=float(m.rgb_std), lib_rows=m.patch_rgb_lib.shape[0], lib_sub=m.patch_rgb_lib[::31, ::16].numpy())
        elif tag == "xyz":
            g.update(mean=float(m.xyz_mean), std=float(m.xyz_std), lib_rows=m.patch_xyz_lib.shape[0], lib_sub=m.patch_xyz_lib[::97, ::16].numpy())
        else:
            main = m.patch_xyz_lib if tag == "mtfi_xyz" else m.patch_rgb_lib
            g.update(mean=float(m.fusion_mean), std=float(m.fusion_std), xyz_mean=float(m.xyz_mean), xyz_std=float(m.xyz_std),
                     rgb_mean=float(m.rgb_mean), rgb_std=float(m.rgb_std), lib_rows=main.shape[0],
                     lib_sub=main[::97 if tag == "mtfi_xyz" else 31, ::16].numpy(), fusion_rows=m.patch_fusion_lib.shape[0],
                     fusion_sub=m.patch_fusion_lib[::97, ::16].numpy())
        out.update({f"{tag}/{k}": v for k, v in g.items()})
        print(tag, "image_preds", g["image_preds"], "s_lib", g["s_lib"][:2])
    np.savez_compressed(os.path.join(HERE, "g11_methods.npz"), **out)
    print("g11_methods.npz", os.path.getsize(os.path.join(HERE, "g11_methods.npz")) // 1024, "KB")


def golden_halluc_depth2():
    """G5b: the reference's HallucinationCrossModalityNetwork with mlp_depth=2 (utils/utils.py:103-115: two chained MlpBlocks
    per direction) -- both generated outputs, the three losses and the l2 loss after each of two Adam steps, on a seeded
    [2, 32, 1536] batch with the synthetic weights of oracle.nets.synth_state_dict("halluc", 53, mlp_depth=2)."""
    from models.hallucination_network import HallucinationCrossModalityNetwork
    import utils.lr_sched as rlr
    sdh = onets.synth_state_dict("halluc", 53, mlp_depth=2)
    net = HallucinationCrossModalityNetwork(_ns(), 768, 768, hidden_ratio=2.5, mlp_depth=2)
    net.load_state_dict(sdh, strict=True)
    samples = torch.randn(2, 32, 1536, generator=torch.Generator().manual_seed(54))
    xyz, rgb = samples[:, :, :768], samples[:, :, 768:]
    out = {}
    with torch.no_grad():
        out["gen_xyz2rgb"] = net.hallucination_generation(xyz_feature=xyz, out_type="rgb").numpy()
        out["gen_rgb2xyz"] = net.hallucination_generation(rgb_feature=rgb, out_type="xyz").numpy()
        for dm in ("l2", "cos_dist", "smooth_l1"):
            a, b = net(xyz, rgb, False, dm)
            out[f"loss_{dm}"] = np.array([a.item(), b.item()])
    opt = torch.optim.Adam(net.parameters(), lr=0.0005)
    sargs = _ns(lr=0.0005, warmup_epochs=1, epochs=10)
    losses = []
    net.train()
    opt.zero_grad()
    for it in range(3):
        rlr.adjust_learning_rate(opt, it / 4 + 0, sargs)
        lx, lr_ = net(xyz, rgb, False, "l2")
        losses.append([lx.item(), lr_.item()])
        (lx + lr_).backward()
        opt.step()
        opt.zero_grad()
    out["train_losses"] = np.array(losses)
    np.savez_compressed(os.path.join(HERE, "g5b_halluc_depth2.npz"), samples_seed=54, weights_seed=53, **out)


def golden_coreset_tf32():
    """G9b: get_coreset_idx_randomp with coreset_dtype='TF32' (features.py:390-391: an fp32 scan -- allow_tf32 touches matrix
    products only), the hard-coded "cuda" redirected to the CPU as for G9."""
    _install_stubs()
    sys.path.insert(0, REF)
    from feature_extractors import features as rfeat
    g = torch.Generator().manual_seed(92)
    z = torch.randn(2500, 768, generator=g)
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: orig_to(self, *[("cpu" if (isinstance(x, str) and x == "cuda") else x) for x in a], **k)
    try:
        fake = _ns(args=_ns(dist_method_coreset="l2"), random_state=0)
        sel = rfeat.Features.get_coreset_idx_randomp(fake, z, n=250, eps=0.9, coreset_dtype="TF32")
    finally:
        torch.Tensor.to = orig_to
    np.savez_compressed(os.path.join(HERE, "g9b_coreset_tf32.npz"), z_seed=92, rows=2500, dim=768, n=250, eps=0.9, random_state=0,
                        idx=sel.numpy().astype(np.int64))


def write_pair_files(root, n=12, class_name="bagel"):
    """The --save_frgb_xyz / --save_rgb_fxyz files of n train samples as THIS package's drop-in writes them
    (DoubleRGBPointFeatures._save_pairs, driven without an extractor: sample i's tensors are filled with i, 100 + i, ...)."""
    from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures
    me = types.SimpleNamespace(args=_ns(save_frgb_xyz=True, save_rgb_fxyz=True, save_path_frgb_xyz=os.path.join(root, "frgb_xyz"),
                                        save_path_rgb_fxyz=os.path.join(root, "rgb_fxyz")),
                               class_name=class_name, ins_id2=0, ins_id3=0,
                               _engine=types.SimpleNamespace(xyz_patch=lambda ex, P: ex))
    for i in range(n):
        sample = (torch.full((1, 3, 224, 224), float(i)), torch.full((1, 3, 224, 224), 100.0 + i))
        DoubleRGBPointFeatures._save_pairs(me, [sample], torch.full((1, 784, 768), 300.0 + i), torch.full((1, 3136, 768), 200.0 + i),
                                           torch.full((1, 3136, 768), 400.0 + i), "train")
    return me


def golden_datasets():
    """G13: the reference's OWN pair datasets (dataset.py:268-362) over files this package's drop-in wrote: per class and data_type the
    length and, per index, which sample's tensors come back in which order (every tensor is filled with a value that names its
    sample and kind).  map_location='cuda' of the feature-to-input class is served from the host here (no GPU in this container)."""
    import tempfile
    import dataset as rds      # the reference's dataset.py
    out = {}
    with tempfile.TemporaryDirectory() as td:
        write_pair_files(td)
        orig = torch.load
        torch.load = lambda f, map_location=None, **kw: orig(f, map_location="cpu", **kw)
        try:
            for cls in ("FeatureToInputPreTrainTensorDataset", "InputToFeaturePreTrainTensorDataset"):
                for dt in ("xyz_frgb", "rgb_fxyz"):
                    ds = getattr(rds, cls)(os.path.join(td, dt.split("_")[1] + "_" + dt.split("_")[0] if dt == "xyz_frgb" else dt, "train"), dt)
                    rows = []
                    for i in range(len(ds)):
                        a, b = ds[i]
                        rows.append([float(a.flatten()[0]), a.dim(), a.shape[0], float(b.flatten()[0]), b.dim(), b.shape[0]])
                    out[f"{cls}.{dt}"] = np.array(rows)
        finally:
            torch.load = orig
    np.savez_compressed(os.path.join(HERE, "g13_datasets.npz"), n=12, **out)
    print({k: v.shape for k, v in out.items()})


def golden_surface():
    """G14: what the reference's scripts that the drop-in does NOT replace (dataset.py, main.py, cmdiad_runner.py,
    hallucination_network_pretrain.py) take from the modules it DOES redirect -- names only, from their import statements (and, for
    `from m import *`, the names of m they use) -- and the reference's own resize_organized_pc (utils/mvtec3d_util.py:14-22) on
    small seeded scans, both output forms."""
    import ast
    import cmdiad_amd
    need = set()
    for f in ("dataset.py", "main.py", "cmdiad_runner.py", "hallucination_network_pretrain.py"):
        tree = ast.parse(open(os.path.join(REF, f)).read())
        used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)} | {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute)}
        for n in ast.walk(tree):
            mods = []
            if isinstance(n, ast.ImportFrom) and n.module in cmdiad_amd._DROPIN:
                for a in n.names:
                    if a.name == "*":
                        mods.append(n.module)
                    else:
                        need.add(f"{n.module}:{a.name}")
            if isinstance(n, ast.ImportFrom) and n.module and any(f"{n.module}.{a.name}" in cmdiad_amd._DROPIN for a in n.names):
                mods += [f"{n.module}.{a.name}" for a in n.names if f"{n.module}.{a.name}" in cmdiad_amd._DROPIN]
            if isinstance(n, ast.Import):
                mods += [a.name for a in n.names if a.name in cmdiad_amd._DROPIN]
            for mod in mods:      # a module object or a star import: every top-level name of the reference's module that the script mentions
                ref_mod = ast.parse(open(os.path.join(REF, mod.replace(".", "/") + ".py")).read())
                defined = {d.name for d in ref_mod.body if isinstance(d, (ast.FunctionDef, ast.ClassDef))}
                need |= {f"{mod}:{name}" for name in defined & used}
    # every top-level function, class and method of the redirected modules with its positional argument names
    sigs = []
    for mod in cmdiad_amd._DROPIN:
        tree = ast.parse(open(os.path.join(REF, mod.replace(".", "/") + ".py")).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                sigs.append(f"{mod}|{node.name}|{','.join(a.arg for a in node.args.posonlyargs + node.args.args)}")
            elif isinstance(node, ast.ClassDef):
                sigs.append(f"{mod}|{node.name}|")
                for sub in node.body:
                    if isinstance(sub, ast.FunctionDef):
                        star = "*" if sub.args.vararg else ""
                        sigs.append(f"{mod}|{node.name}.{sub.name}|{','.join(a.arg for a in sub.args.posonlyargs + sub.args.args)}{star}")
    from utils import mvtec3d_util as rmv
    out = {"names": np.array(sorted(need)), "signatures": np.array(sorted(sigs))}
    rs = np.random.RandomState(14)
    for i, (H, W, h, w) in enumerate([(37, 53, 16, 24), (40, 40, 56, 56), (61, 29, 7, 30), (48, 48, 48, 48)]):
        scan = rs.randn(H, W, 3).astype(np.float32)
        scan[rs.rand(H, W) < 0.3] = 0.0
        out[f"scan_{i}"] = scan
        out[f"size_{i}"] = np.array([h, w])
        out[f"tensor_{i}"] = rmv.resize_organized_pc(scan, target_height=h, target_width=w).numpy()
        out[f"array_{i}"] = rmv.resize_organized_pc(scan, target_height=h, target_width=w, tensor_out=False)
        out[f"depth_{i}"] = np.ascontiguousarray(rmv.organized_pc_to_depth_map(scan))
    np.savez_compressed(os.path.join(HERE, "g14_surface.npz"), **out)
    print(list(out["names"]))


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    if len(sys.argv) > 1 and sys.argv[1] == "g14":  # only the drop-in surface fixture
        return golden_surface()
    if len(sys.argv) > 1 and sys.argv[1] == "g13":  # only the pair-dataset fixture
        return golden_datasets()
    if len(sys.argv) > 1 and sys.argv[1] == "g5b":  # only the mlp_depth = 2 hallucination fixture
        return golden_halluc_depth2()
    if len(sys.argv) > 1 and sys.argv[1] == "g10":  # only the distillation-head fixture
        return golden_heads()
    if len(sys.argv) > 1 and sys.argv[1] == "g11":  # only the method-class fixture
        return golden_methods()
    if len(sys.argv) > 1 and sys.argv[1] == "g12":  # only the head-training fixture
        return golden_heads_train()
    from models import models as rmodels
    from models import pointnet2_utils as rp2
    from models.hallucination_network import HallucinationCrossModalityNetwork
    from feature_extractors import features as rfeat
    from feature_extractors import multiple_features as rmf
    from utils import utils as rutils
    import utils.lr_sched as rlr

    torch.manual_seed(0)
    out = {}

    # ---------------- G1: interpolating_points (pointnet2_utils.py:45-75)
    g = torch.Generator().manual_seed(11)
    pc, _ = rmf.organized_pc_to_unorganized_pc_no_zeros((None, synth_cloud(5, 0.06)))
    xyz1 = pc[:, :, :2000].contiguous()
    cidx, _ = ok.fps(xyz1.permute(0, 2, 1).numpy(), 64)
    xyz2 = xyz1[:, :, torch.from_numpy(cidx[0]).long()]
    feat = torch.randn(1, 16, 64, generator=g)
    res = rp2.interpolating_points(xyz1, xyz2, feat)
    np.savez_compressed(os.path.join(HERE, "g1_interp.npz"), xyz1=xyz1.numpy(), xyz2=xyz2.numpy(),
                        feat=feat.numpy(), out=res.numpy())

    # ---------------- G1b: organized_pc_to_unorganized_pc_no_zeros (multiple_features.py:10-25)
    opc = synth_cloud(7, 0.4)
    pc, nz = rmf.organized_pc_to_unorganized_pc_no_zeros((None, opc))
    np.savez_compressed(os.path.join(HERE, "g1b_unorganize.npz"), seed=7, frac=0.4, n=pc.shape[2],
                        nz_head=nz[:64], nz_tail=nz[-64:], nz_sum=np.int64(nz.sum()),
                        pc_head=pc[0, :, :32].numpy())

    # ---------------- G2: Point-MAE encoder / transformer / full PointTransformer (models.py:183-373)
    sd = onets.synth_state_dict("pointmae", 21)
    pt = rmodels.PointTransformer(group_size=32, num_group=64)
    missing = pt.load_state_dict(sd, strict=True)
    pc, _ = rmf.organized_pc_to_unorganized_pc_no_zeros((None, synth_cloud(9, 0.08)))
    pc = pc[:, :, :3000].contiguous()
    for mode in ("eval", "train"):
        pt.eval() if mode == "eval" else pt.train()
        with torch.no_grad():
            feats, center, ori_idx, center_idx = pt(pc)
            nb, _, _, _ = pt.group_divider(pc.transpose(-1, -2))
            tok = pt.encoder(nb)
        out[f"g2_{mode}"] = dict(feats=feats.numpy(), tokens=tok.numpy())
    np.savez_compressed(os.path.join(HERE, "g2_pointmae.npz"), pc=pc.numpy(), center=center.numpy(),
                        center_idx=center_idx.numpy(), ori_idx=ori_idx.numpy().astype(np.int32),
                        feats_eval=out["g2_eval"]["feats"], tokens_eval=out["g2_eval"]["tokens"],
                        feats_train=out["g2_train"]["feats"], tokens_train=out["g2_train"]["tokens"])

    # ---------------- GV: ViT-B/8 block stack through the reference's in-tree Block (models.py:163-180)
    from functools import partial
    sdv = onets.synth_state_dict("vit", 31)
    blocks = torch.nn.ModuleList([
        rmodels.Block(dim=768, num_heads=12, mlp_ratio=4.0, qkv_bias=True,
                      norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)) for _ in range(12)])
    blocks.load_state_dict({k[len("blocks."):]: v for k, v in sdv.items() if k.startswith("blocks.")})
    blocks.eval()
    g = torch.Generator().manual_seed(32)
    x = torch.randn(1, 785, 768, generator=g)
    with torch.no_grad():
        y = x
        for b in blocks:
            y = b(y)
    np.savez_compressed(os.path.join(HERE, "gv_vit_blocks.npz"), x_seed=32, y_sub=y[0, ::8, ::4].numpy(),
                        y_mean=y.mean().item(), y_std=y.std().item())

    # ---------------- G3: get_xyz_patch (56 and 28) / get_rgb_patch (features.py:160-184)
    fake = _ns(xyz_size=224, average=torch.nn.AvgPool2d(3, stride=1),
               resize28=torch.nn.AdaptiveAvgPool2d((28, 28)), resize56=torch.nn.AdaptiveAvgPool2d((56, 56)))
    opc = synth_cloud(13, 0.5)
    pc, nz = rmf.organized_pc_to_unorganized_pc_no_zeros((None, opc))
    g = torch.Generator().manual_seed(14)
    interp = torch.randn(1, 8, pc.shape[2], generator=g)
    p56 = rfeat.Features.get_xyz_patch(fake, [torch.zeros(1, 8, 4)], interp, nz)
    p28 = rfeat.Features.get_xyz_patch(fake, [torch.zeros(1, 8, 4)], interp, nz, get_2828=True)
    rgbmap = torch.randn(1, 12, 28, 28, generator=g)
    rp, rp2_ = rfeat.Features.get_rgb_patch(fake, [rgbmap])
    np.savez_compressed(os.path.join(HERE, "g3_patch.npz"), cloud_seed=13, frac=0.5, interp_seed=14,
                        n=pc.shape[2], p56=p56.numpy(), p28=p28.numpy(), rgbmap=rgbmap.numpy(),
                        rgb_patch=rp.numpy(), rgb_patch2=rp2_.numpy())

    # ---------------- G4: calculate_dist + compute_single_s_s_map (features.py:186-297)
    class NoBlur:
        def __call__(self, x):
            return x[0]

    g4 = {}
    for tag, (Q, Nb, D, modal) in {"xyz_small": (3136, 2000, 64, "xyz"), "rgb_small": (784, 1500, 64, "rgb"),
                                   "fusion_small": (3136, 1800, 64, "fusion"),
                                   "xyz_fullD": (784, 1200, 768, "xyz")}.items():
        g = torch.Generator().manual_seed({"xyz_small": 41, "rgb_small": 42, "fusion_small": 43, "xyz_fullD": 44}[tag])
        bank = torch.randn(Nb, D, generator=g)
        patch = bank[torch.randint(0, Nb, (Q,), generator=g)] + 0.3 * torch.randn(Q, D, generator=g)
        patch[Q // 3] += 1.5  # a planted anomalous patch
        fs = _ns(args=_ns(dist_method_s="l2"), n_reweight=3, gt_size=224, blur=NoBlur(),
                 patch_xyz_lib=bank, patch_rgb_lib=bank, patch_fusion_lib=bank)
        fs.calculate_dist = lambda a, b, fs=fs: rfeat.Features.calculate_dist(fs, a, b)
        dist = fs.calculate_dist(patch, bank)
        side = int(Q ** 0.5)
        s, s_map = rfeat.Features.compute_single_s_s_map(fs, patch, dist, (side, side), modal=modal)
        mv, mi = torch.min(dist, dim=1)
        g4.update({f"{tag}_seed": {"xyz_small": 41, "rgb_small": 42, "fusion_small": 43, "xyz_fullD": 44}[tag],
                   f"{tag}_shape": np.array([Q, Nb, D]), f"{tag}_min_val": mv.numpy(),
                   f"{tag}_min_idx": mi.numpy().astype(np.int32), f"{tag}_s": s.numpy(),
                   f"{tag}_s_map": s_map.numpy()[:, ::4, ::4]})
    # blur (utils/utils.py:71-83) with the torchvision stub above + real PIL
    g = torch.Generator().manual_seed(45)
    smooth = torch.nn.functional.interpolate(torch.rand(1, 1, 56, 56, generator=g) * 3.0, size=(224, 224),
                                             mode="bilinear")
    blurred = rutils.KNNGaussianBlur(4)(smooth)
    g4.update(blur_seed=45, blur_out=blurred.numpy()[:, ::2, ::2])
    np.savez_compressed(os.path.join(HERE, "g4_score.npz"), **g4)

    # ---------------- G5: hallucination net forward / losses / Adam steps
    sdh = onets.synth_state_dict("halluc", 51)
    args = _ns()
    net = HallucinationCrossModalityNetwork(args, 768, 768, hidden_ratio=2.5, mlp_depth=1)
    net.load_state_dict(sdh, strict=True)
    g = torch.Generator().manual_seed(52)
    samples = torch.randn(2, 64, 1536, generator=g)
    xyz, rgb = samples[:, :, :768], samples[:, :, 768:]
    g5 = {}
    with torch.no_grad():
        g5["gen_xyz2rgb"] = net.hallucination_generation(xyz_feature=xyz, out_type="rgb").numpy()
        g5["gen_rgb2xyz"] = net.hallucination_generation(rgb_feature=rgb, out_type="xyz").numpy()
        for dm in ("l2", "cos_dist", "smooth_l1"):
            a, b = net(xyz, rgb, False, dm)
            g5[f"loss_{dm}"] = np.array([a.item(), b.item()])
    # three Adam steps as hallucination_network_pretrain.py:102-154 (lr 5e-4... warm-up via lr_sched.py)
    opt = torch.optim.Adam(net.parameters(), lr=0.0005)
    sargs = _ns(lr=0.0005, warmup_epochs=1, epochs=10)
    steps_per_epoch = 4
    losses = []
    probes = ["xyz_mlp.mlp_module.0.fc1.weight", "rgb_mlp.mlp_module.0.fc3.bias", "xyz_norm.weight"]
    net.train()
    opt.zero_grad()
    for it in range(3):
        rlr.adjust_learning_rate(opt, it / steps_per_epoch + 0, sargs)
        lx, lr_ = net(xyz, rgb, False, "l2")
        loss = lx + lr_
        losses.append([lx.item(), lr_.item()])
        loss.backward()
        opt.step()
        opt.zero_grad()
        if it in (0, 2):
            for p in probes:
                t = dict(net.named_parameters())[p].detach()
                g5[f"step{it + 1}_{p}"] = (t[:8, :8] if t.dim() == 2 else t[:16]).numpy().copy()
    g5["train_losses"] = np.array(losses)
    np.savez_compressed(os.path.join(HERE, "g5_halluc.npz"), samples_seed=52, **g5)

    # ---------------- G8: seeded default initialisation of the reference's own modules (checksums only)
    g8 = {}
    torch.manual_seed(123)
    pt2 = rmodels.PointTransformer(group_size=128, num_group=1024)
    for k, v in pt2.state_dict().items():
        g8["pm/" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    torch.manual_seed(321)
    hn2 = HallucinationCrossModalityNetwork(_ns(), 768, 768, hidden_ratio=2.5, mlp_depth=1)
    for k, v in hn2.state_dict().items():
        g8["hn/" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    np.savez_compressed(os.path.join(HERE, "g8_init.npz"), **g8)

    # ---------------- G7: calculate_au_pro (utils/au_pro_util.py) on synthetic maps
    from utils.au_pro_util import calculate_au_pro
    rs = np.random.RandomState(7)
    gts, preds = [], []
    for i in range(6):
        gt = np.zeros((64, 64), dtype=int)
        if i % 2 == 0:
            gt[10 + i:22 + i, 8:20] = 1
            gt[40:50, 30 + i:44] = 1
        pr = rs.rand(64, 64) * 0.5 + gt * (0.2 + 0.1 * i) * rs.rand(64, 64)
        gts.append(gt); preds.append(pr)
    au03, _ = calculate_au_pro(gts, preds)
    au001, _ = calculate_au_pro(gts, preds, 0.01)
    np.savez_compressed(os.path.join(HERE, "g7_aupro.npz"), gts=np.array(gts), preds=np.array(preds),
                        au_pro_03=au03, au_pro_001=au001)


    # ---------------- G6: the DINO+Point_MAE protocol through the REFERENCE's own glue
    # (multiple_features.py:800-1015 DoubleRGBPointFeatures; features.py:123-297, 352-358).  The two backbones are
    # replaced by this repo's CPU restatements (timm / pointnet2_ops / knn_cuda are absent), so the
    # fixture pins everything the reference itself owns between the backbone outputs and the final scores: patch
    # extraction, the cross-wired statistics (F5), bank normalisation, cdist + min, the re-weighting, bilinear
    # up-sampling, the 8-bit blur, the lambda weights, the late-fusion bank and the two one-class SVMs.
    from cmdiad_amd.synth import synth_rgb

    class FakeModel(torch.nn.Module):
        def __init__(self, device, rgb_backbone_name, xyz_backbone_name, group_size, num_group):
            super().__init__()
            self.sd_vit = onets.synth_state_dict("vit", 31)
            self.sd_pm = onets.synth_state_dict("pointmae", 21)
            self.G, self.M = num_group, group_size

        def forward(self, rgb, xyz, out_type="rgb+xyz"):
            with torch.no_grad():
                fmap = onets.vit_forward(self.sd_vit, rgb)
                pts = np.ascontiguousarray(xyz[0].T.numpy())[None]
                cidx, cen = ok.fps(pts, self.G)
                idx, nb = ok.knn_group(pts, cen, self.M)
                center = torch.from_numpy(cen)
                tok = onets.pointmae_encoder(self.sd_pm, torch.from_numpy(nb))
                feats = onets.pointmae_transformer(self.sd_pm, tok, center)
            return fmap, feats, center, torch.from_numpy(idx), torch.from_numpy(cidx)

    rfeat.Model = FakeModel
    args = _ns(rgb_backbone_name="vit_base_patch8_224_dino", xyz_backbone_name="Point_MAE", group_size=128, num_group=1024,
               rgb_size=224, xyz_size=224, gt_size=224, f_coreset=0.25, coreset_eps=0.9, coreset_dtype="FP16",
               random_state=0, dist_method_s="l2", dist_method_coreset="l2", main_modality="", use_hn=False,
               use_hn_conv=False, use_hn_from_rgb_mlp=False, use_hn_from_rgb_conv=False, use_hrnet=False, use_uff=False,
               use_depth=False, fusion_module_path="", ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0,
               xyz_smap_lambda=1.0, rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0,
               save_feature_for_fusion=False, save_frgb_xyz=False, save_rgb_fxyz=False, save_seg_results=False,
               save_raw_results=False, save_path="", save_path_frgb_xyz="", save_path_rgb_fxyz="", experiment_note="", c_hrnet=0)
    G6_TRAIN, G6_TEST = (201, 202, 203), (211, 212)
    sample = lambda sd: (synth_rgb(sd), synth_cloud(sd, 0.45, texture=0.004), synth_cloud(sd, 0.45, texture=0.004))
    m = rmf.DoubleRGBPointFeatures(args)
    for sd in G6_TRAIN:
        m.add_sample_to_mem_bank(sample(sd), class_name="synthetic")
    # f_coreset < 1 as in real runs: with the whole train set in the bank every late-fusion query would match itself and
    # the scores would be fp32 cancellation noise.  run_coreset reaches get_coreset_idx_randomp, which hard-codes
    # .to("cuda") (features.py:397-399): that one device string is redirected to the CPU for the call.
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: orig_to(self, *[("cpu" if (isinstance(x, str) and x == "cuda") else x) for x in a], **k)
    picked = []
    inner = m.get_coreset_idx_randomp
    m.get_coreset_idx_randomp = lambda *a, **k: (picked.append(inner(*a, **k)), picked[-1])[1]  # record both selections
    try:
        m.run_coreset()
    finally:
        torch.Tensor.to = orig_to
    for sd in G6_TRAIN:
        m.add_sample_to_late_fusion_mem_bank(sample(sd))
    s_lib = torch.cat(m.s_lib, 0).clone()
    s_map_lib = torch.cat(m.s_map_lib, 0).clone()
    m.run_late_fusion()
    for sd in G6_TEST:
        m.predict(sample(sd), torch.zeros(1, 224, 224), 0, ["x.png"])
    g6 = dict(train_seeds=np.array(G6_TRAIN), test_seeds=np.array(G6_TEST), frac=0.45, texture=0.004, f_coreset=0.25, random_state=0,
              xyz_mean=float(m.xyz_mean), xyz_std=float(m.xyz_std), rgb_mean=float(m.rgb_mean), rgb_std=float(m.rgb_std),
              xyz_lib_rows=m.patch_xyz_lib.shape[0], rgb_lib_rows=m.patch_rgb_lib.shape[0],
              xyz_coreset_idx=picked[0].numpy().astype(np.int32), rgb_coreset_idx=picked[1].numpy().astype(np.int32),
              xyz_lib_sub=m.patch_xyz_lib[::97, ::16].numpy(), rgb_lib_sub=m.patch_rgb_lib[::31, ::16].numpy(),
              s_lib=s_lib.numpy(), s_map_lib_sub=s_map_lib[::53].numpy(),
              detect_coef=m.detect_fuser.coef_, detect_offset=m.detect_fuser.offset_,
              seg_coef=m.seg_fuser.coef_, seg_offset=m.seg_fuser.offset_,
              image_preds=np.array(m.image_preds).reshape(-1), pred_maps_sub=np.array(m.predictions)[:, ::4, ::4])
    np.savez_compressed(os.path.join(HERE, "g6_protocol.npz"), **g6)

    # ---------------- G9: get_coreset_idx_randomp (features.py:360-425), which hard-codes .to("cuda"): run here with
    # that one device string redirected to the CPU (generator-only patch; the arithmetic is the reference's)
    g = torch.Generator().manual_seed(91)
    z = torch.randn(2500, 768, generator=g)
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: orig_to(self, *[("cpu" if (isinstance(x, str) and x == "cuda") else x) for x in a], **k)
    try:
        fake = _ns(args=_ns(dist_method_coreset="l2"), random_state=0)
        sel = rfeat.Features.get_coreset_idx_randomp(fake, z, n=250, eps=0.9, coreset_dtype="FP16")
    finally:
        torch.Tensor.to = orig_to
    np.savez_compressed(os.path.join(HERE, "g9_coreset.npz"), z_seed=91, rows=2500, dim=768, n=250, eps=0.9, random_state=0,
                        idx=sel.numpy().astype(np.int64))

    golden_coreset_tf32()
    golden_heads()
    golden_methods()

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
