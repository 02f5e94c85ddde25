"""cProfile of the host side of (a) the end-to-end evaluation at 224^2 / batch 64 and (b) the serving call at batch 1."""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "salient-object-detection_amd"))
import bench  # noqa: E402
from selfmask_amd import datasets as DS  # noqa: E402
from selfmask_amd.evaluator import Evaluator  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w = bench.Workload(dev, 16, 224, 64)
    root = tempfile.mkdtemp(prefix="sm_hp_")
    try:
        DS.write_synthetic_dataset(root, "duts", 768, seed=7)
        sub, di, _, dg, _ = DS.LAYOUTS["duts"]
        n = 768 * 12
        for i in range(768, n):
            for d_, ext in ((di, "jpg"), (dg, "png")):
                os.symlink(os.path.join(root, sub, d_, f"{i % 768:05d}.{ext}"), os.path.join(root, sub, d_, f"{i:05d}.{ext}"))
        ev = Evaluator(network=w.model, dir_dataset=root)
        ev.device = dev
        ev("duts", dir_ckpt=os.path.join(root, "ck"), img_size=224, batch_size=64, device=dev, streams=3)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t = time.perf_counter()
        pr.enable()
        ev("duts", dir_ckpt=os.path.join(root, "ck"), img_size=224, batch_size=64, device=dev, streams=3)
        torch.cuda.synchronize()
        pr.disable()
        dt = time.perf_counter() - t
        print(f"== end to end, 224^2, batch 64: {n / dt:.0f} images/s ({n} images, {dt / (n / 64) * 1e3:.2f} ms per batch)")
        pstats.Stats(pr).sort_stats("tottime").print_stats(16)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    from argparse import Namespace
    from selfmask_amd.inference import SelfMaskInference
    inf = SelfMaskInference(None, Namespace(), device=dev, model=w.model)
    rgb = np.random.default_rng(5).integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    for _ in range(20):
        inf.predict_tensors(rgb)
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    for _ in range(300):
        inf.predict_tensors(rgb)
    pr.disable()
    dt = time.perf_counter() - t
    print(f"== serving, batch 1: {dt / 300 * 1e3:.3f} ms per request under the profiler")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
