"""`cpu_baseline`: the CPU oracle pipeline timed on the box's host cores.  This module and tests/ are the only importers of
oracle/ (the checker); the product package never does."""
import time

from .common import N_POINTS, RGB_ROWS, XYZ_ROWS

def cpu_baseline(n_images=10, warm=3):
    """The CPU oracle pipeline (oracle/pipeline.py, kind 'port': the reference's own torch-CPU composition + the C restatement
    of FPS / kNN) on a bounded sample of the same workload, on this box's host cores: at the thread count that is fastest
    here (`value`), at the reference's default --cpu_core_num 6 (main.py:149) and at all cores."""
    import torch
    from cmdiad_amd.models.models import PointTransformer, VisionTransformer
    from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n, synth_rgb
    from oracle.pipeline import CpuDoubleRGBPoint, CpuExtractor
    torch.manual_seed(0)
    sd_vit = {k: v.detach() for k, v in VisionTransformer().state_dict().items()}
    sd_pm = {k: v.detach() for k, v in PointTransformer().state_dict().items()}
    cpu = CpuDoubleRGBPoint(CpuExtractor(sd_vit, sd_pm))
    cpu.set_banks(synth_bank(XYZ_ROWS, 768, 4321), synth_bank(RGB_ROWS, 768, 4322), 0.0, 1.0, 0.0, 1.0)
    all_threads = torch.get_num_threads()

    def timed(threads, n, w):
        torch.set_num_threads(threads)
        for i in range(w):
            cpu.predict(synth_rgb(i), synth_cloud_fixed_n(1000 + i, N_POINTS))
        cpu.ex.timing.clear(); cpu.timing.clear()
        t0 = time.perf_counter()
        for i in range(n):
            cpu.predict(synth_rgb(w + i), synth_cloud_fixed_n(1000 + w + i, N_POINTS))
        dt = time.perf_counter() - t0
        stages = {k: round(v / n, 4) for k, v in {**cpu.ex.timing, **cpu.timing}.items()}
        return n / dt, stages

    # torch's intra-op pool oversubscribes badly beyond ~32 threads on these shapes (128-thread MI355X host, round 1: 0.18
    # images/s at 128 threads, 0.47 at 32, 0.35 at 6), so the best setting is measured, not assumed
    best_t = min(32, all_threads)
    v_best, stages = timed(best_t, n_images, warm)
    v_six, _ = timed(min(6, all_threads), max(3, n_images // 3), 1)
    v_all, _ = (v_best, None) if all_threads == best_t else timed(all_threads, max(3, n_images // 3), 1)
    torch.set_num_threads(all_threads)
    return dict(value=round(v_best, 4), unit="images/s", cores=best_t, kind="port",
                sample=f"{n_images} images after {warm} warm-up, B=1, fp32, torch {torch.__version__} CPU ({best_t} intra-op threads: "
                       f"the fastest setting on this {all_threads}-thread host) + C oracle for FPS/kNN, same synthetic inputs and "
                       f"bagel-sized banks",
                at_reference_default_6_threads=round(v_six, 4), at_all_threads={"threads": all_threads, "value": round(v_all, 4)},
                seconds_per_image_by_stage=stages)
