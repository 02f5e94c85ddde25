"""Minimal test-mode dataset reader with the reference's semantics (datasets/base_dataset.py:228-256, duts.py:29-30,
108-147, ecssd.py:17-18, dut_omron.py:17-18, utils/misc.py:78-107) and a synthetic dataset writer.

Out of scope as a subsystem (I/O plumbing); kept to what the evaluator needs: sorted file
listing of the three benchmark layouts, RGB decode, ``ToTensor`` (/255) + ``Normalize`` (ImageNet mean/std), GT in
mode "L" binarised with ``m > 0`` when its max exceeds 1, native resolution (the reference's test mode) or a fixed
S x S bilinear resize (batched mode, as app.py:198-205 does for its 224^2 path).
"""
import os
from glob import glob
from os.path import join
from typing import List, Optional, Tuple

import numpy as np
import torch
from PIL import Image

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)

# dataset_name -> (sub-directory, image dir, image glob, GT dir, GT glob)
LAYOUTS = {
    "ecssd": ("ECSSD", "images", "*.jpg", "ground_truth_mask", "*.png"),
    "duts": ("DUTS", "DUTS-TE-Image", "*.jpg", "DUTS-TE-Mask", "*.png"),
    "dut_omron": ("DUTS-OMRON", "DUT-OMRON-image", "*.jpg", "pixelwiseGT-new-PNG", "*.png"),
}


def probe_size(path: str) -> Tuple[int, int]:
    """(H, W) of an image file from its header: the JPEG frame header (SOFn) or the PNG IHDR read directly - a few reads per file against
    PIL's plugin machinery (45 us per file: 0.09 s of a 0.6-s run over 2 048 files) - and PIL itself for anything else or anything odd.
    Like ``Image.open(path).size``, EXIF orientation is not applied."""
    try:
        with open(path, "rb") as f:
            head = f.read(26)
            if head[:8] == b"\x89PNG\r\n\x1a\n" and head[12:16] == b"IHDR":
                return int.from_bytes(head[20:24], "big"), int.from_bytes(head[16:20], "big")
            if head[:2] == b"\xff\xd8":
                f.seek(2)
                while True:
                    b = f.read(1)
                    if not b:
                        break
                    if b != b"\xff":
                        continue
                    m = f.read(1)
                    while m == b"\xff":  # fill bytes
                        m = f.read(1)
                    if not m:
                        break
                    code = m[0]
                    if code in (0x01, 0xD8) or 0xD0 <= code <= 0xD7:  # stand-alone markers
                        continue
                    if code in (0xD9, 0xDA):  # end of image / start of scan before any frame header
                        break
                    seg = f.read(2)
                    if len(seg) < 2:
                        break
                    length = int.from_bytes(seg, "big")
                    if 0xC0 <= code <= 0xCF and code not in (0xC4, 0xC8, 0xCC):  # SOF0..SOF15 without DHT, JPG, DAC
                        fr = f.read(5)
                        if len(fr) == 5:
                            return int.from_bytes(fr[1:3], "big"), int.from_bytes(fr[3:5], "big")
                        break
                    f.seek(length - 2, 1)
    except OSError:
        pass
    with Image.open(path) as im:
        w, h = im.size
    return h, w


class SaliencyTestDataset:
    def __init__(self, dir_dataset: str, dataset_name: str, eval_img_size: Optional[int] = None):
        if dataset_name not in LAYOUTS:
            raise ValueError(f"{dataset_name} not in {sorted(LAYOUTS)}")
        sub, di, gi, dg, gg = LAYOUTS[dataset_name]
        self.name = dataset_name
        self.p_imgs: List[str] = sorted(glob(join(dir_dataset, sub, di, gi)))
        self.p_gts: List[str] = sorted(glob(join(dir_dataset, sub, dg, gg)))
        assert len(self.p_imgs) == len(self.p_gts), f"{len(self.p_imgs)} != {len(self.p_gts)}"
        self.img_size = eval_img_size

    def __len__(self) -> int:
        return len(self.p_imgs)

    def image_size(self, ind: int) -> Tuple[int, int]:
        """(H, W) from the file header only (PIL opens lazily): the native-resolution evaluator plans its token-grid buckets
        from these before anything is decoded."""
        return probe_size(self.p_imgs[ind])

    def __getitem__(self, ind: int) -> dict:
        image = Image.open(self.p_imgs[ind]).convert("RGB")
        if self.img_size is not None:
            image = image.resize((self.img_size, self.img_size), Image.BILINEAR)
        x = np.asarray(image, np.float32) / np.float32(255.0)  # TF.to_tensor
        x = (x - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)  # TF.normalize
        m = np.asarray(Image.open(self.p_gts[ind]).convert("L"), np.int64)
        if m.max() > 1.0:
            m = m > 0
        return {"x": torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))),
                "m": torch.from_numpy(np.ascontiguousarray(m.astype(np.uint8))),
                "p_img": self.p_imgs[ind], "filename": os.path.basename(self.p_imgs[ind])}


def get_dataset(dir_dataset: str, dataset_name: str, mode: str = "test", eval_img_size: Optional[int] = None, **kw):
    """utils/misc.py:43-151 for the three saliency benchmarks in test mode."""
    assert mode == "test", "only the evaluation path is implemented"
    return SaliencyTestDataset(dir_dataset, dataset_name, eval_img_size)


def synthetic_scene(rng: np.random.Generator, h: int, w: int) -> Tuple[np.ndarray, np.ndarray]:
    """One random-ellipse scene: (RGB uint8 (h, w, 3), ground truth bool (h, w)) - smooth background, one or two salient ellipses
    of another colour, sensor noise."""
    yy, xx = np.mgrid[:h, :w]
    img = np.empty((h, w, 3), np.float32)
    for c in range(3):
        img[..., c] = 110 + 60 * np.sin(xx / rng.uniform(15, 60) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(15, 60))
    gt = np.zeros((h, w), bool)
    for _ in range(int(rng.integers(1, 3))):
        cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
        ry, rx = rng.uniform(0.1, 0.3) * h, rng.uniform(0.1, 0.3) * w
        e = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1
        gt |= e
        img[e] += rng.uniform(-90, 90, size=3).astype(np.float32)
    img = np.clip(img + rng.standard_normal(img.shape) * 8, 0, 255).astype(np.uint8)
    return img, gt


def write_synthetic_dataset(dir_dataset: str, dataset_name: str, n_images: int, seed: int = 7,
                            size_range: Tuple[int, int] = (300, 400)) -> None:
    """Random-ellipse scenes in the reference's directory layout (there are no real datasets offline)."""
    sub, di, gi, dg, gg = LAYOUTS[dataset_name]
    os.makedirs(join(dir_dataset, sub, di), exist_ok=True)
    os.makedirs(join(dir_dataset, sub, dg), exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(seed))
    for i in range(n_images):
        h, w = (int(v) for v in rng.integers(size_range[0], size_range[1] + 1, size=2))
        img, gt = synthetic_scene(rng, h, w)
        Image.fromarray(img).save(join(dir_dataset, sub, di, f"{i:05d}.jpg"), quality=92)
        Image.fromarray((gt * 255).astype(np.uint8)).save(join(dir_dataset, sub, dg, f"{i:05d}.png"))
