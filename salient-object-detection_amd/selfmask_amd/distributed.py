"""Image-sharded evaluation: rank r takes images r, r+W, ...; ONE all-gather of per-image result rows; every rank
then reduces in global index order with the reference's sequential AverageMeter arithmetic, so the W-rank result is
bit-identical to the 1-rank result.  The reference has no distributed code at all."""
from typing import List, Optional, Sequence

import numpy as np
import torch

ROW = 16  # 14 metric values + picked query + upper-bound query
HEADER = ("iou,pixel_acc,f_score,f_max,f_mean,mae,s_measure,miou_ub,pixel_acc_ub,f_score_ub,f_max_ub,f_mean_ub,"
          "mae_ub,s_measure_ub\n")  # evaluator.pyc@L276 (note miou_ub vs the dict key iou_ub)
KEYS = ("iou", "pixel_accuarcy", "f_score", "f_max", "f_mean", "mae", "s_measure")  # dict keys, evaluator.pyc@L294-308


def rank_cores(local_rank: int, local_world: int, cores: Sequence[int]) -> List[int]:
    """This rank's share of a set of cores: a contiguous block (neighbouring cores share caches / a NUMA node)."""
    cores = sorted(cores)
    per = max(1, len(cores) // max(1, local_world))
    lo = min(local_rank * per, max(0, len(cores) - per))
    return cores[lo:lo + per]


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_topology(sysfs: str = "/sys") -> List[dict]:
    """The node's GPUs in HIP device order with the host cores next to each: [{"numa_node": int, "cpus": [...]}, ...].
    KFD lists every compute node under class/kfd/kfd/topology/nodes/<i>/properties; the ones with SIMDs are GPUs, in the order
    the runtime enumerates them, and `drm_render_minor` names their DRM render node, whose PCI device directory holds
    `numa_node` and `local_cpulist`.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (index lists) are applied.  Empty when the
    tree is absent (no GPU, another OS): the caller then falls back to contiguous blocks."""
    import os
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    gpus = []
    try:
        nodes = sorted((d for d in os.listdir(base) if d.isdigit()), key=int)
    except OSError:
        return []
    for d in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(base, d, "properties")) if " " in line.strip())
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        dev = os.path.join(sysfs, "class", "drm", f"renderD{int(props.get('drm_render_minor', '-1'))}", "device")
        try:
            numa = int(open(os.path.join(dev, "numa_node")).read())
            cpus = _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
        except (OSError, ValueError):
            numa, cpus = -1, []
        gpus.append({"numa_node": numa, "cpus": cpus})
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):  # ROCR filters first, HIP indexes what is left
        sel = os.environ.get(var)
        if sel:
            try:
                gpus = [gpus[int(i)] for i in sel.split(",") if i.strip() != ""]
            except (ValueError, IndexError):
                return []  # UUID lists or stale indices: do not guess
    return gpus


def numa_rank_cores(local_rank: int, local_world: int, affinity: Sequence[int], topology: Sequence[dict]) -> List[int]:
    """Rank r drives GPU r: its decode workers belong on the cores of that GPU's NUMA node.  The ranks whose GPUs
    hang off the same node split that node's cores (within the affinity mask) into contiguous blocks, in rank order.  Falls back to
    the plain contiguous split when the topology is unknown, a GPU reports no node (-1), or a node has fewer usable cores than ranks."""
    plain = rank_cores(local_rank, local_world, affinity)
    if len(topology) < local_world or any(t["numa_node"] < 0 or not t["cpus"] for t in topology[:local_world]):
        return plain
    node = topology[local_rank]["numa_node"]
    peers = [r for r in range(local_world) if topology[r]["numa_node"] == node]
    usable = sorted(set(affinity) & set(topology[local_rank]["cpus"]))
    if len(usable) < len(peers):
        return plain
    return rank_cores(peers.index(local_rank), len(peers), usable)


def pin_rank_cores(sysfs: str = "/sys") -> List[int]:
    """One process per GPU, eight of them on one host: give each rank its own block of the host's cores (its decode workers
    inherit it) instead of 8 x N threads fighting over all of them - the cores of the NUMA node its GPU hangs off when sysfs says
    which (gpu_numa_topology), a contiguous block of the affinity mask otherwise.  Reads LOCAL_RANK / LOCAL_WORLD_SIZE
    (torch.distributed.run); a no-op for a single rank.  Sets SM_RANK_CORES_PINNED so that decode_pool.default_workers does not
    divide again."""
    import os
    lw, lr = int(os.environ.get("LOCAL_WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    try:
        cur = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    if lw <= 1 or os.environ.get("SM_RANK_CORES_PINNED") == "1":
        return cur
    mine = numa_rank_cores(lr, lw, cur, gpu_numa_topology(sysfs))
    os.sched_setaffinity(0, mine)
    os.environ["SM_RANK_CORES_PINNED"] = "1"
    return mine


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    return list(range(rank, n_items, world_size))


class TorchDistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, or "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        # concatenated along dim 0 (the layout both the nccl/RCCL and the gloo backends accept), viewed as stacked
        src = t.contiguous()
        if src.is_cuda and self.dist.get_backend(self.group) == "gloo":  # gloo gathers host tensors (CPU tests, one-GPU rehearsals)
            src = src.cpu()
        out = torch.empty((self.world_size * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        self.dist.all_gather_into_tensor(out, src, group=self.group)
        return out.view((self.world_size,) + tuple(t.shape)).to(t.device)


class SingleComm:
    rank, world_size = 0, 1

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        return t.unsqueeze(0)


def gather_rows(rows_local: torch.Tensor, idx_local: Sequence[int], n_total: int, comm) -> np.ndarray:
    """rows_local (n_local, 16) on any device -> (n_total, 16) float32 numpy in global index order on every rank.
    Payload per rank: ceil(n/W) x 17 fp32 (global index + row), padded with index -1."""
    per = -(-n_total // comm.world_size)
    buf = torch.full((per, ROW + 1), -1.0, dtype=torch.float32, device=rows_local.device)
    n = len(idx_local)
    if n:
        buf[:n, 0] = torch.tensor(list(idx_local), dtype=torch.float32, device=rows_local.device)
        buf[:n, 1:] = rows_local
    allb = comm.all_gather(buf).reshape(-1, ROW + 1).cpu().numpy()
    out = np.full((n_total, ROW), np.nan, np.float32)
    seen = np.zeros(n_total, bool)
    for r in allb:
        i = int(r[0])
        if i >= 0:
            assert not seen[i], f"image {i} evaluated twice"
            out[i], seen[i] = r[1:], True
    assert seen.all(), f"{int((~seen).sum())} images missing after the gather"
    return out


def gather_bytes(payload: bytes, comm, device="cpu") -> List[bytes]:
    """One variable-length byte string per rank -> the list of all ranks' strings, in rank order, on every rank: an all-gather of the
    lengths, then one of the payloads padded to the longest (device tensors under RCCL, host tensors under gloo)."""
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    lens = comm.all_gather(n).reshape(-1).tolist()
    buf = torch.zeros(max(max(lens), 1), dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    allb = comm.all_gather(buf).cpu().numpy()
    return [allb[r, :lens[r]].tobytes() for r in range(comm.world_size)]


def gather_dicts(local: dict, comm, device="cpu") -> dict:
    """{key: JSON-able value} of every rank's shard -> the union on every rank (pseudo-mask generation, BASELINE.json configs[4]: the
    file list is sharded like the evaluator's images, the per-file run-length codes are the only thing exchanged).  A key two ranks
    both hold is an error: shards are disjoint."""
    import json
    out: dict = {}
    for r, b in enumerate(gather_bytes(json.dumps(local, separators=(",", ":")).encode(), comm, device)):
        part = json.loads(b.decode()) if b else {}
        dup = set(part) & set(out)
        assert not dup, f"rank {r} repeats {sorted(dup)[:3]}"
        out.update(part)
    return out


def average_rows(rows: np.ndarray) -> dict:
    """metrics/average_meter.py:12-16 over images in index order: float32 running sums for the tensor-derived
    values, Python-float (fp64) sums for the S-measure (SMeasure returns ``Q.item()``), evaluator.pyc@L55-99."""
    out = {}
    for k in range(14):
        col = rows[:, k]
        if k % 7 == 6:  # s_measure / s_measure_ub
            s = 0.0
            for v in col:
                s += float(v)
            avg = s / len(col)
        else:
            s = np.float32(0)
            for v in col:
                s = np.float32(s + np.float32(v))
            avg = float(np.float32(s / np.float32(len(col))))
        out[KEYS[k % 7] + ("_ub" if k >= 7 else "")] = avg
    return out
