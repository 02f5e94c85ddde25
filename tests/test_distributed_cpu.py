"""N>1 path on CPU: world_size-2 gloo processes shard the image list, all-gather their per-image rows and must
reproduce the single-process averages bit for bit (sequential float32 AverageMeter arithmetic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import evaluator_oracle as E
from selfmask_amd import distributed as D


def _rows(n, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    r = rng.random((n, 16)).astype(np.float32)
    r[:, 14:] = rng.integers(0, 20, size=(n, 2))
    return r


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 7, 16, 5019):
        for w in (1, 2, 3, 8):
            got = sorted(i for r in range(w) for i in D.shard_indices(n, r, w))
            assert got == list(range(n))


def test_average_rows_is_the_reference_average_meter():
    rows = _rows(37)
    res = D.average_rows(rows)
    for k in range(14):
        m = E.AverageMeter()
        for v in rows[:, k]:
            # tensor-derived values reach the meter as numpy 0-d float32; S-measure as a Python float
            m.update(val=float(v) if k % 7 == 6 else np.float32(v), n=1)
        key = D.KEYS[k % 7] + ("_ub" if k >= 7 else "")
        assert res[key] == float(m.avg), (key, res[key], m.avg)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = _rows(n)
    mine = D.shard_indices(n, rank, world)
    comm = D.TorchDistComm()
    full = D.gather_rows(torch.from_numpy(rows[mine]), mine, n, comm)
    res = D.average_rows(full)
    np.save(os.path.join(outdir, f"rows_{rank}.npy"), full)
    np.save(os.path.join(outdir, f"avg_{rank}.npy"), np.array([res[k] for k in sorted(res)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [16, 17, 1])
def test_two_rank_gloo_gather_equals_single_rank(tmp_path, n):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    rows = _rows(n)
    single = D.average_rows(D.gather_rows(torch.from_numpy(rows), list(range(n)), n, D.SingleComm()))
    ref = np.array([single[k] for k in sorted(single)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rows_{r}.npy"), rows)
        assert np.array_equal(np.load(tmp_path / f"avg_{r}.npy"), ref)  # bit-identical on every rank


def test_dataset_reader_semantics(tmp_path):
    from PIL import Image
    from selfmask_amd import datasets as DS
    DS.write_synthetic_dataset(str(tmp_path), "duts", 3, seed=1, size_range=(40, 60))
    DS.write_synthetic_dataset(str(tmp_path), "dut_omron", 2, seed=2, size_range=(40, 60))
    ds = DS.get_dataset(str(tmp_path), "duts")
    assert len(ds) == 3 and ds.p_imgs == sorted(ds.p_imgs)
    it = ds[1]
    img = np.asarray(Image.open(it["p_img"]).convert("RGB"), np.float32)
    ref = (img / 255.0 - np.array(DS.MEAN, np.float32)) / np.array(DS.STD, np.float32)
    assert np.allclose(it["x"].numpy(), ref.transpose(2, 0, 1), atol=1e-6)
    assert it["m"].dtype == torch.uint8 and set(np.unique(it["m"].numpy())) <= {0, 1} and it["m"].shape == img.shape[:2]
    assert DS.get_dataset(str(tmp_path), "dut_omron", eval_img_size=32)[0]["x"].shape == (3, 32, 32)
    with pytest.raises(ValueError):
        DS.get_dataset(str(tmp_path), "cub")


def test_rank_cores_partition_the_host():
    from selfmask_amd.distributed import rank_cores
    cores = list(range(3, 131))  # 128 cores, not starting at 0
    blocks = [rank_cores(r, 8, cores) for r in range(8)]
    assert all(len(b) == 16 for b in blocks) and sorted(sum(blocks, [])) == cores   # disjoint, complete, contiguous
    assert all(b == list(range(b[0], b[0] + 16)) for b in blocks)
    assert rank_cores(0, 1, cores) == cores and len(rank_cores(5, 8, list(range(6)))) == 1   # fewer cores than ranks: one each


def _fake_sysfs(root, gpus, cpu_nodes=2):
    """a sysfs tree as KFD / DRM lay it out: `cpu_nodes` CPU nodes first, then one node per GPU = (numa_node, local_cpulist)"""
    import os
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for i in range(cpu_nodes):
        os.makedirs(os.path.join(nodes, str(i)))
        open(os.path.join(nodes, str(i), "properties"), "w").write("cpu_cores_count 48\nsimd_count 0\ndrm_render_minor 0\n")
    for g, (numa, cpulist) in enumerate(gpus):
        d = os.path.join(nodes, str(cpu_nodes + g))
        os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        dev = os.path.join(root, "class", "drm", f"renderD{128 + g}", "device")
        os.makedirs(dev)
        open(os.path.join(dev, "numa_node"), "w").write(f"{numa}\n")
        open(os.path.join(dev, "local_cpulist"), "w").write(cpulist + "\n")


def test_rank_cores_follow_the_numa_node_of_the_ranks_gpu(tmp_path, monkeypatch):
    """'each rank needs its own decode workers pinned to its NUMA node' (VERDICT r3 #12).  A two-socket host, GPUs
    0-3 on node 0 (cores 0-47, 96-143), GPUs 4-7 on node 1 (cores 48-95, 144-191)."""
    from selfmask_amd.distributed import gpu_numa_topology, numa_rank_cores, rank_cores
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    root = str(tmp_path / "sys")
    _fake_sysfs(root, [(0, "0-47,96-143")] * 4 + [(1, "48-95,144-191")] * 4)
    topo = gpu_numa_topology(root)
    assert [t["numa_node"] for t in topo] == [0, 0, 0, 0, 1, 1, 1, 1] and topo[5]["cpus"][:2] == [48, 49] and len(topo[0]["cpus"]) == 96
    affinity = list(range(192))
    blocks = [numa_rank_cores(r, 8, affinity, topo) for r in range(8)]
    assert sorted(c for b in blocks for c in b) == affinity  # a partition of the host
    for r, b in enumerate(blocks):
        assert set(b) <= set(topo[r]["cpus"]) and len(b) == 24  # every rank on its GPU's node
    # a container that may only use cores 0-63: node 0 keeps 48 of them for its four ranks, node 1 the other 16
    blocks = [numa_rank_cores(r, 8, list(range(64)), topo) for r in range(8)]
    assert [len(b) for b in blocks] == [12] * 4 + [4] * 4 and set(blocks[6]) <= set(range(48, 64))
    # visible-device lists re-index the GPUs
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6,1")
    assert [t["numa_node"] for t in gpu_numa_topology(root)] == [1, 0]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # unknown topology, a GPU without a node, or fewer usable cores than ranks: the plain contiguous split
    assert numa_rank_cores(3, 8, affinity, []) == rank_cores(3, 8, affinity)
    assert gpu_numa_topology(str(tmp_path / "nothing")) == []
    bad = [dict(t) for t in topo]
    bad[2]["numa_node"] = -1
    assert numa_rank_cores(2, 8, affinity, bad) == rank_cores(2, 8, affinity)
    assert numa_rank_cores(7, 8, list(range(3)), topo) == rank_cores(7, 8, list(range(3)))


def test_pin_rank_cores_uses_the_topology(tmp_path, monkeypatch):
    import os
    from selfmask_amd import distributed as D
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no CPU affinity on this platform")
    cur = sorted(os.sched_getaffinity(0))
    if len(cur) < 4:
        pytest.skip("needs four cores")
    half = len(cur) // 2
    lo, hi = cur[:half], cur[half:]
    root = str(tmp_path / "sys")
    as_list = lambda cs: ",".join(str(c) for c in cs)
    _fake_sysfs(root, [(0, as_list(hi)), (1, as_list(lo))])  # GPU 0 next to the UPPER half: a contiguous split would get this wrong
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.delenv("SM_RANK_CORES_PINNED", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    try:
        assert D.pin_rank_cores(root) == hi and sorted(os.sched_getaffinity(0)) == hi
    finally:
        os.sched_setaffinity(0, cur)
        os.environ.pop("SM_RANK_CORES_PINNED", None)


def _dict_worker(rank, world, port, n, outdir):
    import json
    import torch.distributed as dist
    from selfmask_amd.mask_generator import rle_encode
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = D.TorchDistComm()
    local = {f"img_{i}.png": rle_encode(_mask(i)) for i in D.shard_indices(n, rank, world)}
    with open(os.path.join(outdir, f"merged_{rank}.json"), "w") as f:
        json.dump(D.gather_dicts(local, comm), f, sort_keys=True)
    dist.destroy_process_group()


def _mask(i):
    rng = np.random.Generator(np.random.PCG64(i))
    return (rng.random((7 + i % 5, 9 + i % 3)) < 0.4).astype(np.uint8)


@pytest.mark.parametrize("n", [9, 2, 1, 0])
def test_two_rank_gloo_gather_of_encoded_pseudo_masks(tmp_path, n):
    """configs[4] across ranks: the file list sharded rank-strided, every rank ends with every file's run-length code - ragged payloads,
    a rank with nothing (n < world), nothing at all."""
    import json
    from selfmask_amd.mask_generator import rle_decode
    world = 2
    mp.spawn(_dict_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    merged = [json.load(open(tmp_path / f"merged_{r}.json")) for r in range(world)]
    assert merged[0] == merged[1] and sorted(merged[0]) == sorted(f"img_{i}.png" for i in range(n))
    for i in range(n):
        assert np.array_equal(rle_decode(merged[0][f"img_{i}.png"]), _mask(i))
    assert D.gather_dicts({"a": 1}, D.SingleComm()) == {"a": 1}
