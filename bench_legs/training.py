"""configs[2]: the distillation training steps (`train_step`: feature-to-feature MLP; `conv_head_train_step`: HRNet head)."""
import os
import time

from .common import PEAK_BF16_TFLOPS

def train_step_leg(dev, steps=50, warm=10):
    """BASELINE configs[2]: one FtoF distillation training step = both directions forward + loss + backward + Adam on a
    [32, 3136, 1536] feature batch (xyz first, rgb second), N(0,1), seed 3407 (hallucination_network_pretrain.py:53,102-159), lr
    schedule per iteration (utils/lr_sched.py:4-17), l2 loss; 7.99 TFLOP per step (3 x forward, both directions, 100 352
    tokens).  The batch is resident in HBM (tools/train_bench.py also times the FeatureRing-fed loop)."""
    import types
    import torch
    from cmdiad_amd import train
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.utils import lr_sched
    torch.manual_seed(3407)
    net = HallucinationCrossModalityNetwork(None, 768, 768).to(dev)
    opt = train.FusedAdam(net.parameters(), lr=5e-4)
    sched = types.SimpleNamespace(lr=5e-4, warmup_epochs=10, epochs=100)
    x = torch.randn(32, 3136, 1536, generator=torch.Generator(device=dev).manual_seed(3407), device=dev)
    losses = []

    def step(it):
        lr_sched.adjust_learning_rate(opt, it / 100.0, sched)
        lx, lr_ = net(x[:, :, :768], x[:, :, 768:], False, "l2")
        opt.zero_grad(set_to_none=True)
        (lx + lr_).backward()
        opt.step()
        return lx, lr_

    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        lx, lr_ = step(warm + i)
        if i in (0, steps - 1):
            losses.append((lx, lr_))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    l0, l1 = [float(a.item() + b.item()) for a, b in losses]
    assert l1 == l1 and l1 < l0, (l0, l1)            # finite, and the optimiser is descending
    return dict(what="configs[2]: HallucinationCrossModality feature-to-feature distillation training step (forward + l2 loss + "
                     "backward, both directions, + Adam) on [32, 3136, 1536] synthetic features resident in HBM",
                ms_per_step=round(dt * 1e3, 3), steps_per_s=round(1.0 / dt, 2), tflop_per_step=7.99,
                achieved_TFLOPs=round(7.99 / dt, 1), frac_of_mfma_peak=round(7.99 / dt / PEAK_BF16_TFLOPS, 4),
                steps=steps, warmup=warm, loss_first_timed=round(l0, 2), loss_last_timed=round(l1, 2),
                tokens_per_s=round(32 * 3136 / dt, 0))


def conv_head_train_leg(dev, batch=8, steps=6, warm=2):
    """one training step of the convolutional FtoF head (HallucinationCrossModalityConv: per direction conv3x3 ->
    batch-statistics BatchNorm -> ReLU three times + conv3x3, hallucination_network.py:72-147) -- both directions, forward + l2 loss +
    backward + Adam -- on the hand-written path of cmdiad_amd/conv_train.py.  FLOPs: 2 towers x (4 forward + 3 data-gradient + 4
    weight-gradient convolutions) x 2 M 768 (9 768), M = batch x 3136 positions."""
    import torch
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityConv
    torch.manual_seed(3407)
    net = HallucinationCrossModalityConv(None, 768, 768).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(3407)
    a, b = torch.randn(batch, 3136, 768, generator=g, device=dev), torch.randn(batch, 3136, 768, generator=g, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        lx, lr_ = net(a, b, False, "l2")
        (lx + lr_).backward()
        opt.step()
        return lx, lr_

    first = None
    for i in range(warm):
        lx, lr_ = step()
        first = first if first is not None else float(lx.detach() + lr_.detach())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lx, lr_ = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    last = float(lx.detach() + lr_.detach())
    assert last == last and last < first, (first, last)
    tflop = 2 * 11 * 2.0 * batch * 3136 * 768 * 9 * 768 / 1e12
    return dict(what="row f4: HallucinationCrossModalityConv training step (both directions: forward, l2 loss, backward, Adam), "
                     "hand-written HIP forward + backward (cmdiad_amd/conv_train.py), batch-statistics BatchNorm",
                batch=batch, ms_per_step=round(dt * 1e3, 2), tflop_per_step=round(tflop, 2), achieved_TFLOPs=round(tflop / dt, 1),
                steps=steps, warmup=warm, loss_first=round(first, 1), loss_last=round(last, 1))
