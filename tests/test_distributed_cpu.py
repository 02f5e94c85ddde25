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
