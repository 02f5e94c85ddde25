"""ORACLE (test infrastructure, NOT product code): CPU restatement of the fast bilateral solver refinement.

Follows /root/reference/bilateral_solver.py (Barron & Poole's solver as carried by LOST/TokenCut): grid construction
:41-85, splat/slice/blur :87-104, bistochastize :107-118, solve :127-149, bilateral_solver_output :152-193.  The
linear algebra is restated with explicit numpy index arrays (no scipy.sparse); the PCG loop restates
scipy.sparse.linalg.cg (SciPy 1.15 `_isolve/iterative.py:cg`; the reference pins scipy==1.7.3 whose `tol` is the new
`rtol`, atol=0 -> stop when ||r|| < tol*||b||).  scipy.ndimage.binary_fill_holes / label are the reference's own call
sites (:184-185) and are called as such.

Parity status: PINNED by tests/golden/bilateral.npz (outputs of the real reference module on three synthetic
images, made by oracle/gen_golden.py): vertex and blur-nnz counts exact, soft output to 1e-9, binary mask identical.
"""
import math

import numpy as np
from scipy import ndimage

RGB_TO_YUV = np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]])
YUV_OFFSET = np.array([0.0, 128.0, 128.0])
MAX_VAL = 255.0


def _fma(a, b, c):
    """fp64 fused multiply-add, element-wise (exact product in long double, one rounding)."""
    return (a.astype(np.longdouble) * np.longdouble(b) + c.astype(np.longdouble)).astype(np.float64)


def rgb2yuv(im: np.ndarray) -> np.ndarray:
    """bilateral_solver.py:21-22.  np.tensordot lowers to dgemm with K=3, whose kernel accumulates
    fma(B, m2, fma(G, m1, R*m0)); the order decides the luma/chroma bin of exactly-grey pixels (Y = v*1.0 -> v-ulp)."""
    imf = im.astype(np.float64)
    out = np.empty(im.shape[:2] + (3,), np.float64)
    for c in range(3):
        acc = imf[..., 0] * RGB_TO_YUV[c, 0]
        acc = _fma(imf[..., 1], RGB_TO_YUV[c, 1], acc)
        acc = _fma(imf[..., 2], RGB_TO_YUV[c, 2], acc)
        out[..., c] = acc + YUV_OFFSET[c]
    return out


class Grid:
    """BilateralGrid (:40-104) with explicit neighbour index arrays instead of CSR matrices."""

    def __init__(self, im, sigma_spatial=16, sigma_luma=16, sigma_chroma=8):
        yuv = rgb2yuv(im)
        iy, ix = np.mgrid[:im.shape[0], :im.shape[1]]
        coords = np.dstack(((ix / sigma_spatial).astype(int), (iy / sigma_spatial).astype(int),
                            (yuv[..., 0] / sigma_luma).astype(int), (yuv[..., 1:] / sigma_chroma).astype(int)))
        flat = coords.reshape(-1, 5)
        self.npixels, self.dim = flat.shape
        hash_vec = MAX_VAL ** np.arange(5)
        hashed = flat.astype(np.float64) @ hash_vec
        uniq, uidx, self.idx = np.unique(hashed, return_index=True, return_inverse=True)
        ucoords = flat[uidx]
        self.nvertices = len(uniq)
        # neighbour tables: nb[d][0] = vertex at coord-1 along d (or -1), nb[d][1] = at coord+1
        self.nb = np.full((5, 2, self.nvertices), -1, np.int64)
        self.blur_nnz = []
        for d in range(5):
            nnz = 0
            for s, off in enumerate((-1, 1)):
                vec = np.zeros(5)
                vec[d] = off
                nh = (ucoords + vec).astype(np.float64) @ hash_vec
                loc = np.clip(np.searchsorted(uniq, nh), 0, len(uniq) - 1)
                ok = uniq[loc] == nh
                self.nb[d, s, ok] = loc[ok]
                nnz += int(ok.sum())
            self.blur_nnz.append(nnz)

    def splat(self, x):  # S.dot(x): per vertex, sum over its pixels in ascending pixel order
        out = np.zeros(self.nvertices)
        np.add.at(out, self.idx, x)  # np.add.at accumulates sequentially in index order
        return out

    def slice(self, y):
        return y[self.idx]

    def blur(self, x):
        out = 2 * self.dim * x
        for d in range(5):
            t = np.zeros_like(x)
            for s in range(2):  # CSR row order: the -1 neighbour (smaller vertex id) first
                ok = self.nb[d, s] >= 0
                t[ok] = t[ok] + x[self.nb[d, s][ok]]
            out = out + t
        return out


def bistochastize(grid: Grid, maxiter=10):
    """:107-118"""
    m = grid.splat(np.ones(grid.npixels))
    n = np.ones(grid.nvertices)
    for _ in range(maxiter):
        n = np.sqrt(n * m / grid.blur(n))
    m = n * grid.blur(n)
    return n, m


def _matvec(grid: Grid, n, diag, p, lam):
    """A.dot(p) with A = lam*(Dm - Dn blur(Dn)) + diag(S w) assembled like scipy: per row, ascending column order
    (neighbours with a smaller vertex id, the diagonal, neighbours with a larger id); off-diagonal entries are
    -(lam * n_v * n_j)."""
    V = grid.nvertices
    cols = np.concatenate([grid.nb.reshape(10, V), np.arange(V)[None]], 0)  # (11, V)
    vals = np.concatenate([-(lam * (n[None] * n[np.clip(grid.nb.reshape(10, V), 0, V - 1)])), diag[None]], 0)
    vals[:10][grid.nb.reshape(10, V) < 0] = 0.0
    key = np.where(cols >= 0, cols, V + 1)
    order = np.argsort(key, axis=0, kind="stable")
    cols_s = np.take_along_axis(cols, order, 0)
    vals_s = np.take_along_axis(vals, order, 0)
    out = np.zeros(V)
    for k in range(11):
        ok = cols_s[k] >= 0
        out[ok] = out[ok] + vals_s[k][ok] * p[cols_s[k][ok]]
    return out


def solve(grid: Grid, n, m, t, w, lam=256.0, diag_min=1e-5, maxiter=25, tol=1e-5):
    """BilateralSolver.solve :127-149 + scipy cg."""
    w_splat = grid.splat(w)
    diag = lam * (m - n * (2 * grid.dim * n)) + w_splat
    b = grid.splat(t * w)
    Minv = 1.0 / np.maximum(diag, diag_min)
    x = b / w_splat
    bnrm2 = np.linalg.norm(b)
    atol = tol * bnrm2
    if bnrm2 == 0:
        return grid.slice(b)
    r = b - _matvec(grid, n, diag, x, lam) if x.any() else b.copy()
    rho_prev, p = None, None
    for it in range(maxiter):
        if np.linalg.norm(r) < atol:
            break
        z = Minv * r
        rho_cur = np.dot(r, z)
        if it > 0:
            p *= rho_cur / rho_prev
            p += z
        else:
            p = z.copy()
        q = _matvec(grid, n, diag, p, lam)
        alpha = rho_cur / np.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho_cur
    return grid.slice(x)


def postprocess(soft: np.ndarray):
    """:184-192: threshold, fill holes, 4-connected labelling, keep the SECOND largest label (background included)."""
    h, w = soft.shape
    binary = ndimage.binary_fill_holes(soft > 0.5)
    labeled, nr = ndimage.label(binary)
    sizes = [np.sum(labeled == i) for i in range(nr + 1)]
    order = np.argsort(sizes)
    try:
        return labeled == order[-2]
    except IndexError:
        return np.ones((h, w), dtype=bool)


def bilateral_solver_output(img: np.ndarray, target: np.ndarray, sigma_spatial=16, sigma_luma=16, sigma_chroma=8):
    """:152-193 (img: (H,W,3) uint8 RGB array instead of a PIL image)."""
    h, w = target.shape
    grid = Grid(img, sigma_spatial, sigma_luma, sigma_chroma)
    n, m = bistochastize(grid)
    t = target.reshape(-1).astype(np.float64)
    c = np.ones(h * w) * 0.999
    soft = solve(grid, n, m, t, c).reshape(h, w)
    return soft, postprocess(soft), grid
