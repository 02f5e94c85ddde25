"""Host half of one evaluation sample and the worker process that runs it.

``decode_item`` is the reference's per-sample host work (datasets/base_dataset.py:248-255, duts.py:123,144): RGB decode of the
image, GT decode in mode "L" binarised with ``m > 0`` when its maximum exceeds 1.  This module imports only numpy and Pillow - it
is also the entry point of the decode WORKER PROCESSES of ``selfmask_amd.decode_pool`` (``python sm_decode_worker.py``), which
must start fast and must never touch the GPU.

Why processes: Pillow releases the GIL only inside libjpeg / zlib; everything around it (the Python-level header parsing of
``Image.open``, ``convert``, ``np.asarray``, ``max``, ``astype``) holds it - about half of the 1.7 ms a 350 x 350 sample costs.
Threads therefore top out near 1 / (GIL-held time): measured 595 images/s on ONE thread, 414 on four and 482 on eight threads
of the same 8-core host; 16 threads of a GPU box reach 1.6-2.0k images/s where 16 cores could decode 9k.

Worker protocol: one request per line on stdin, one tab-separated reply per line on stdout.  A request is tab separated when no field
holds a tab, a newline or a backslash (every ordinary path: 0.5 us to build), and a JSON array behind a leading "J" otherwise (paths
may hold any character):
    request   <image path> <GT path or -> <shm file> <rgb offset> <rgb capacity> <gt offset> <gt capacity>
              J[image path, GT path or null, shm file, rgb offset, rgb capacity, gt offset, gt capacity]
    reply     ok <H> <W> <gtH> <gtW>      pixels written to the shared segment: (H, W, 3) uint8 at rgb offset, (gtH, gtW) uint8
                                           {0, 1} at gt offset (gtH = gtW = 0 without a GT)
              big <H> <W> <gtH> <gtW>     a capacity is too small: nothing written, the parent decodes this sample itself
              err <message>
    request   J["drop", shm file, ...]    the segments are gone (their loader finished): unmap them;  reply  dropped <count>
A worker also unmaps every segment whose file has disappeared whenever it meets a new one, so an unlinked ring of slots never
outlives the next loader even without the message.
"""
import json
import mmap
import os
import sys

import numpy as np
from PIL import Image


def decode_item(p_img: str, p_gt=None):
    """Host part of one sample: RGB decode (+ GT decode and binarisation, base_dataset.py:248-255 / duts.py:123,144)."""
    rgb = np.asarray(Image.open(p_img).convert("RGB"), np.uint8)
    m = None
    if p_gt is not None:
        m = np.asarray(Image.open(p_gt).convert("L"))
        if m.max() > 1:
            m = m > 0
        m = np.ascontiguousarray(m.astype(np.uint8))
    return rgb, m


def _unmap(maps: dict, path: str) -> bool:
    mm = maps.pop(path, None)
    if mm is None:
        return False
    try:
        mm.close()
    except BufferError:  # a numpy view still alive (cannot happen between requests); the mapping goes with it
        pass
    return True


def _serve() -> None:
    maps = {}
    out = sys.stdout
    for line in sys.stdin:
        try:
            if line.startswith("J["):  # (a plain path that begins like that is sent as JSON by the parent)
                req = json.loads(line[1:])
            else:
                req = line.rstrip("\n").split("\t")
                req[1] = None if req[1] == "-" else req[1]
                req[3:] = [int(v) for v in req[3:]]
            if req and req[0] == "drop" and not (len(req) == 7 and isinstance(req[3], int)):
                reply = f"dropped\t{sum(_unmap(maps, f) for f in req[1:])}"
            else:
                p_img, p_gt, shm, ro, rc, go, gc = req
                rgb, m = decode_item(p_img, p_gt)
                h, w = rgb.shape[:2]
                gh, gw = (m.shape if m is not None else (0, 0))
                if rgb.size > rc or (m is not None and m.size > gc):
                    reply = f"big\t{h}\t{w}\t{gh}\t{gw}"
                else:
                    mm = maps.get(shm)
                    if mm is None:
                        for stale in [f for f in maps if not os.path.exists(f)]:
                            _unmap(maps, stale)
                        with open(shm, "r+b") as f:  # the segment's file under /dev/shm (no resource tracker involved)
                            mm = maps[shm] = mmap.mmap(f.fileno(), 0)
                    dst = np.frombuffer(mm, np.uint8, rgb.size, ro)
                    dst[:] = rgb.reshape(-1)
                    del dst
                    if m is not None:
                        dst = np.frombuffer(mm, np.uint8, m.size, go)
                        dst[:] = m.reshape(-1)
                        del dst
                    reply = f"ok\t{h}\t{w}\t{gh}\t{gw}"
        except Exception as e:  # noqa: BLE001 - reported to the parent, which raises
            reply = "err\t" + repr(e).replace("\n", " ").replace("\t", " ")
        out.write(reply + "\n")
        out.flush()


if __name__ == "__main__":
    _serve()
