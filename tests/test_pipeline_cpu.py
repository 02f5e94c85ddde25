"""Host side of the input pipeline (selfmask_amd/pipeline.py) pinned against Pillow and against the reference's
ToTensor + Normalize expressions - no GPU needed: the device kernels apply exactly these tables in int32 / by look-up."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "salient-object-detection_amd"))

from selfmask_amd import pipeline as P  # noqa: E402
from selfmask_amd.datasets import MEAN, STD  # noqa: E402


@pytest.mark.parametrize("h,w,S", [(300, 400, 224), (371, 262, 224), (224, 224, 224), (97, 61, 224), (400, 300, 384),
                                   (1000, 333, 224), (225, 223, 224), (30, 500, 64)])
def test_fixed_point_resize_is_pillow_bit_for_bit(h, w, S):
    rng = np.random.Generator(np.random.PCG64(h * 1000 + w))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    img[: h // 3] = (img[: h // 3] // 64) * 64          # flat areas, saturated edges
    img[-3:, :, 0], img[:, -2:, 1] = 255, 0
    ref = np.asarray(Image.fromarray(img).resize((S, S), Image.BILINEAR))
    got = P.resize_reference_numpy(img, S)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} bytes differ (max {int(np.abs(got.astype(int) - ref).max())})"


def test_tap_tables_shape_and_normalisation():
    b, k, ks = P.pil_resize_coeffs(400, 224)
    assert ks == 5 and b.shape == (224, 2) and k.shape == (224, 5)
    assert (np.abs(k.sum(1) - (1 << P.PRECISION_BITS)) <= 3).all()          # taps sum to 1.0 in 22-bit fixed point
    assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= 400).all()
    b1, k1, _ = P.pil_resize_coeffs(224, 224)                                 # identity resize: one tap of 1.0
    assert (k1[:, 0] == 1 << P.PRECISION_BITS).all() and (b1[:, 0] == np.arange(224)).all()


def test_normalisation_table_is_totensor_then_normalize():
    lut = P.normalize_lut()
    v = torch.arange(256, dtype=torch.uint8)
    for c in range(3):
        t = v.to(torch.float32).div(255)                                      # TF.to_tensor
        t = (t - torch.tensor(MEAN[c])) / torch.tensor(STD[c])                # TF.normalize: sub_ then div_
        assert np.array_equal(lut[c * 256:(c + 1) * 256], t.numpy())


def test_pack_images_layout():
    imgs = [np.full((5, 7, 3), 9, np.uint8), np.full((4, 7, 3), 3, np.uint8)]
    pixels, coef, descr, max_h, max_px, out_elems = P.pack_images(imgs, 16)
    assert max_h == 5 and max_px == 35 and out_elems == 3 * (35 + 28)
    d = np.frombuffer(descr.numpy().tobytes(), dtype=np.dtype([("off", "<i8"), ("out_off", "<i8"), ("H", "<i4"), ("W", "<i4"),
                                                               ("cx", "<i4"), ("cy", "<i4"), ("ksx", "<i4"), ("ksy", "<i4")]))
    assert list(d["H"]) == [5, 4] and list(d["W"]) == [7, 7] and d["off"][1] % 16 == 0 and d["out_off"][1] == 105
    assert d["cx"][0] == d["cx"][1]                                           # equal widths share one tap table
    assert pixels[int(d["off"][1])] == 3


def test_prefetching_loader_order(tmp_path):
    from selfmask_amd import datasets as DS
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 7, seed=3, size_range=(40, 60))
    ds = DS.get_dataset(str(tmp_path), "ecssd")
    seen = []
    for rgbs, gts, idx in P.PrefetchingLoader(ds, range(len(ds)), batch_size=3, workers=2, depth=2):
        assert len(rgbs) == len(gts) == len(idx)
        for rgb, gt, i in zip(rgbs, gts, idx):
            item = ds[i]
            assert np.array_equal(gt, item["m"].numpy()) and rgb.shape[:2] == gt.shape and set(np.unique(gt)) <= {0, 1}
        seen += idx
    assert seen == list(range(7))


def test_native_buckets_partition_by_token_grid():
    from selfmask_amd.pipeline import native_buckets
    rng = np.random.Generator(np.random.PCG64(3))
    sizes = [(int(h), int(w)) for h, w in rng.integers(90, 200, size=(200, 2))]
    for patch, mb in ((16, 8), (8, 5), (16, 1)):
        plan = native_buckets(sizes, patch, mb)
        flat = [i for b in plan for i in b]
        assert sorted(flat) == list(range(len(sizes)))            # every image exactly once
        for b in plan:
            assert 1 <= len(b) <= mb and b == sorted(b)          # dataset order kept inside a batch
            grids = {(-(-sizes[i][0] // patch), -(-sizes[i][1] // patch)) for i in b}
            assert len(grids) == 1                               # one token grid per batch
    assert native_buckets([], 16, 4) == []


def test_decode_pool_processes_write_the_same_bytes_as_decode_item(tmp_path):
    """Worker processes + shared-memory slots (decode_pool.py) against the in-process decode_item: same pixels, same GT bytes;
    a sample beyond its window falls back to the in-process decode; a missing file surfaces as an error; no /dev/shm leftovers."""
    import glob
    from selfmask_amd import datasets as DS
    from selfmask_amd.decode_pool import BatchSlots, DecodePool
    from selfmask_amd.pipeline import PrefetchingLoader, decode_item
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 21, seed=3, size_range=(40, 90))
    ds = DS.get_dataset(str(tmp_path), "ecssd")
    before = set(glob.glob("/dev/shm/sm_decode_*"))
    seen = []
    for mode in ("process", "thread"):
        for rgbs, gts, idx in PrefetchingLoader(ds, range(len(ds)), 8, workers=3, depth=2, decode=mode):
            for r, g, i in zip(rgbs, gts, idx):
                rr, gg = decode_item(ds.p_imgs[i], ds.p_gts[i])
                assert np.array_equal(r, rr) and np.array_equal(g, gg) and g.dtype == np.uint8 and set(np.unique(g)) <= {0, 1}
                seen.append(i)
    assert sorted(seen) == sorted(list(range(len(ds))) * 2)
    pool, slots = DecodePool(2), BatchSlots(1, 4, max_side=64)   # 64 x 64 windows: the larger images come back "big"
    try:
        paths = [(ds.p_imgs[i], ds.p_gts[i]) for i in range(4)]
        rgbs, gts = pool.decode_batch(slots, 0, paths)()
        for (pi, pg), r, g in zip(paths, rgbs, gts):
            rr, gg = decode_item(pi, pg)
            assert np.array_equal(r, rr) and np.array_equal(g, gg)
        with pytest.raises(RuntimeError, match="decode worker"):
            pool.decode_batch(slots, 0, [(str(tmp_path / "missing.jpg"), None)])()
        rgbs, gts = pool.decode_batch(slots, 0, [(ds.p_imgs[0], None)])()   # the pool survives an error; GT-less samples
        assert gts == [None] and np.array_equal(rgbs[0], decode_item(ds.p_imgs[0], None)[0])
    finally:
        pool.close()
        slots.close()
    assert set(glob.glob("/dev/shm/sm_decode_*")) == before


def test_workers_keep_no_mapping_of_a_finished_loader_and_paths_may_hold_any_character(tmp_path):
    """every PrefetchingLoader makes its own ring of /dev/shm slots and unlinks it at the end; the shared
    workers used to keep those unlinked segments mapped for good (tmpfs pages + address space, +7.5 MB per loader).  Now the
    loader tells them to drop the ring (DecodePool.drop), and a worker also unmaps vanished segments when it meets a new one.
    Requests are JSON lines: a path with a tab and a newline decodes like any other.  ``list(loader)`` holds owned arrays."""
    import os
    import shutil
    from selfmask_amd import datasets as DS
    from selfmask_amd.decode_pool import shared_pool
    from selfmask_amd.pipeline import PrefetchingLoader, decode_item
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 9, seed=5, size_range=(40, 70))
    ds = DS.get_dataset(str(tmp_path), "ecssd")
    odd = str(tmp_path / "odd\tname\nwith breaks.jpg")
    shutil.copy(ds.p_imgs[0], odd)
    ds.p_imgs[0] = odd
    os.makedirs(tmp_path / "J[x", exist_ok=True)
    for i, name in ((1, str(tmp_path / "Jpeg_like_name.jpg")), (2, str(tmp_path / "J[x" / "y.jpg"))):  # names that start like a JSON request
        shutil.copy(ds.p_imgs[i], name)
        ds.p_imgs[i] = name
    pool = shared_pool(2)

    def deleted_maps():
        n = 0
        for p in pool._procs:
            with open(f"/proc/{p.pid}/maps") as f:
                n += sum(1 for line in f if "sm_decode_" in line and "(deleted)" in line)
        return n

    for _ in range(3):
        got = list(PrefetchingLoader(ds, range(len(ds)), 4, workers=2, depth=1))
        assert deleted_maps() == 0
    # batches of a finished iteration are still intact (slot k % 2 was reused twice since batch 0 was yielded)
    for rgbs, gts, idx in got:
        for r, g, i in zip(rgbs, gts, idx):
            rr, gg = decode_item(ds.p_imgs[i], ds.p_gts[i])
            assert np.array_equal(r, rr) and np.array_equal(g, gg) and r.flags.owndata
    assert pool.drop(["/dev/shm/never_mapped"]) == 0


def test_loader_falls_back_to_threads_without_shared_memory(tmp_path, monkeypatch):
    import warnings
    from selfmask_amd import datasets as DS
    from selfmask_amd import decode_pool
    from selfmask_amd.pipeline import PrefetchingLoader, decode_item
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 5, seed=6, size_range=(40, 60))
    ds = DS.get_dataset(str(tmp_path), "ecssd")

    def no_shm(*a, **k):
        raise OSError(28, "No space left on device")
    monkeypatch.setattr(decode_pool.BatchSlots, "__init__", no_shm)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = list(PrefetchingLoader(ds, range(len(ds)), 2, workers=2, depth=1))
    assert any("decoding on threads" in str(x.message) for x in w)
    assert sum(len(b[0]) for b in out) == 5 and np.array_equal(out[0][0][0], decode_item(ds.p_imgs[0], ds.p_gts[0])[0])


def test_probe_size_reads_what_pil_reads(tmp_path):
    """datasets.probe_size (JPEG frame header / PNG IHDR read directly) against Image.open(...).size: baseline, progressive and
    optimised JPEGs, with an EXIF-sized APP1 segment in front, grayscale, PNG (RGB, palette, 16-bit), and a format it hands to PIL."""
    from PIL import Image
    from selfmask_amd.datasets import probe_size
    rng = np.random.Generator(np.random.PCG64(5))
    files = []
    for i, (h, w) in enumerate([(300, 400), (37, 53), (1, 1), (641, 17)]):
        img = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
        for tag, kw in (("base", {}), ("prog", {"progressive": True}), ("opt", {"optimize": True, "quality": 60})):
            p = str(tmp_path / f"{i}_{tag}.jpg")
            img.save(p, **kw)
            files.append(p)
        p = str(tmp_path / f"{i}_exif.jpg")
        img.save(p, exif=b"Exif\x00\x00" + bytes(40000))  # a long APP1 segment before the frame header
        files.append(p)
        img.convert("L").save(str(tmp_path / f"{i}_gray.jpg"))
        files.append(str(tmp_path / f"{i}_gray.jpg"))
        for tag, im2 in (("rgb", img), ("pal", img.convert("P")), ("i16", Image.fromarray(rng.integers(0, 65535, size=(h, w), dtype=np.uint16)))):
            p = str(tmp_path / f"{i}_{tag}.png")
            im2.save(p)
            files.append(p)
        p = str(tmp_path / f"{i}.bmp")
        img.save(p)
        files.append(p)
    for p in files:
        with Image.open(p) as im:
            assert probe_size(p) == (im.size[1], im.size[0]), p
