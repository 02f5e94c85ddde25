"""cProfile of the native-resolution evaluation in token-grid buckets (the reference's own operating point, batched)."""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

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
    root = tempfile.mkdtemp(prefix="sm_np_")
    try:
        DS.write_synthetic_dataset(root, "duts", 768, seed=7)
        sub, di, _, dg, _ = DS.LAYOUTS["duts"]
        for i in range(768, 3072):
            for d_, ext in ((di, "jpg"), (dg, "png")):
                os.symlink(os.path.join(root, sub, d_, f"{i % 768:05d}.{ext}"), os.path.join(root, sub, d_, f"{i:05d}.{ext}"))
        ev = Evaluator(network=w.model, dir_dataset=root)
        ev.device = dev
        for bs in (16, 1):
            ev("duts", dir_ckpt=os.path.join(root, "ck"), batch_size=bs, device=dev, streams=3)
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            t = time.perf_counter()
            pr.enable()
            ev("duts", dir_ckpt=os.path.join(root, "ck"), batch_size=bs, device=dev, streams=3)
            torch.cuda.synchronize()
            pr.disable()
            dt = time.perf_counter() - t
            print(f"== batch_size={bs}: {3072 / dt:.0f} images/s, graphs {ev.graph_stats}")
            pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
