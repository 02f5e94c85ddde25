#!/bin/bash
# pseudo-mask chain by batch size and batches in flight: the eigen-solver runs ONE workgroup per image, so its launch time barely
# moves with the batch, and it leaves most CUs to another batch's encoder
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r4
python - > gpurun_out/r4/pseudo_batches.log 2>&1 <<'PY'
import json, torch, bench
dev = torch.device("cuda:0")
for B in (8, 16, 32, 64, 128, 256):
    for st in (2, 3, 4):
        r = bench.pseudo_masks_leg(dev, st, B=B, steps=12, warmup=3, cpu=False)
        ph = {k: v["ms_per_call"] for k, v in r["clusterer_phases_ms_per_batch"].items()}
        print(f"batch {B} x {st} in flight: chain {r['value']} images/s ({r['ms_per_step']} ms/step), one stream {r['one_stream_images_per_sec']}; "
              f"encoder {r['encoder_ms_per_batch']} ms; clusterer {r['spectral_cluster_ms_per_batch']} ms = {r['spectral_cluster_images_per_sec']} "
              f"images/s; phases {json.dumps(ph)}; matvecs/image {r['eigensolver']['block_matvecs_mean']}, converged "
              f"{r['eigensolver']['converged']}/{B}", flush=True)
PY
cat gpurun_out/r4/pseudo_batches.log
