import io, time, sys
import numpy as np
from PIL import Image
sys.path.insert(0, "/root/repo/salient-object-detection_amd")
rng = np.random.Generator(np.random.PCG64(3))
def scene(h, w):
    yy, xx = np.mgrid[:h, :w]
    img = np.empty((h, w, 3), np.float32)
    for c in range(3):
        img[..., c] = 110 + 60 * np.sin(xx / rng.uniform(15, 60) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(15, 60))
    gt = (((yy - h/2) / (h*.2)) ** 2 + ((xx - w/2) / (w*.25)) ** 2) <= 1
    img[gt] += rng.uniform(-90, 90, size=3).astype(np.float32)
    img = np.clip(img + rng.standard_normal(img.shape) * 8, 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(img).save(b, format="JPEG", quality=92)
    g = io.BytesIO(); Image.fromarray((gt * 255).astype(np.uint8)).save(g, format="PNG")
    return b.getvalue(), g.getvalue()
items = [scene(int(rng.integers(300, 401)), int(rng.integers(300, 401))) for _ in range(64)]
def t(f, n=5):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best / len(items) * 1e6
def full():
    for j, g in items: np.asarray(Image.open(io.BytesIO(j)).convert("RGB"), np.uint8)
def dc_only():
    for j, g in items:
        im = Image.open(io.BytesIO(j)); im.draft("RGB", (im.size[0] // 8, im.size[1] // 8)); np.asarray(im.convert("RGB"), np.uint8)
def header():
    for j, g in items: Image.open(io.BytesIO(j)).size
def png():
    for j, g in items:
        m = np.asarray(Image.open(io.BytesIO(g)).convert("L"))
        if m.max() > 1: m = m > 0
        np.ascontiguousarray(m.astype(np.uint8))
print(f"JPEG full decode        {t(full):7.0f} us per image (mean {np.mean([len(j) for j,_ in items])/1e3:.0f} kB, 300-400 px, q92)")
print(f"JPEG at scale 1/8       {t(dc_only):7.0f} us  (entropy decode of every coefficient + DC-only IDCT: what stays on the host)")
print(f"JPEG header only        {t(header):7.0f} us  (Image.open: Python-level parsing)")
print(f"PNG ground truth        {t(png):7.0f} us  (inflate + unfilter + binarise)")
