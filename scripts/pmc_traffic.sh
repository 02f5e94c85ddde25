#!/bin/bash
# HBM traffic per kernel launch of the bench's forward (guide's recipe: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes,
# no tracing domains; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 on gfx950) -> profiles-ready JSON with the hash of the
# kernel sources it ran on (bench.py quotes roofline.traffic only when that hash matches).
# usage (on the GPU box): bash scripts/pmc_traffic.sh  ->  gpurun_out/${TAG:-r04}_pmc_traffic.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_$C -- python3 bench.py --quick --steps 4 --warmup 2 --streams 1 --no-graph \
      > gpurun_out/pmc_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
from bench import source_hash
def load(c):
    f = sorted(glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sm::", "").strip()
            agg[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {"source_hash": source_hash(), "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --quick --steps 4 --warmup 2 --streams 1 --no-graph",
       "formula": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports half of a wide coalesced read, KB units)",
       "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    f, w = fe.get(k, (0.0, 0))[0], wr.get(k, (0.0, 0))[0]
    out["kernels"][k] = {"launches_sampled": fe.get(k, (0, 0))[1], "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
                         "hbm_bytes_per_launch": round((2 * f + w) * 1024)}
json.dump(out, open("gpurun_out/%s_pmc_traffic.json" % os.environ.get("TAG", "r04"), "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(f'{v["hbm_bytes_per_launch"] / 1e6:9.1f} MB/launch  x{v["launches_sampled"]:5d}  {k}')
PY
