"""End-to-end leg of bench.py alone, for a few decode-worker counts (SM_DECODE_WORKERS): python scripts/e2e_workers.py 12 13 14 15"""
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "salient-object-detection_amd"))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w = bench.Workload(dev, 16, 224, 64)
    for n in [int(v) for v in sys.argv[1:]] or [15]:
        os.environ["SM_DECODE_WORKERS"] = str(n)
        r = bench.end_to_end(w.model, dev, 16, 224, 64, 3)
        print(json.dumps({"workers": n, "end_to_end": r["end_to_end_images_per_sec"], "decode_only": r["host_decode_only_images_per_sec"],
                          "native_b1": r["native_resolution"]["batch1_images_per_sec"],
                          "native_b16": r["native_resolution"]["bucketed_batch16_images_per_sec"]}), flush=True)


if __name__ == "__main__":
    main()
