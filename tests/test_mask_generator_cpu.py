"""Host half of the MaskGenerator mirror (selfmask_amd/mask_generator.py): the COCO uncompressed run-length form and the constructor's
contract; the device half runs in tests/test_hip_spectral.py."""
import numpy as np
import pytest

from selfmask_amd.mask_generator import MaskGenerator, rle_decode, rle_encode


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_rle_round_trip_and_column_major_order(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    m = (rng.random((17 + seed, 23)) > (0.5 if seed else 2.0)).astype(np.uint8)  # seed 0: all zeros
    if seed == 3:
        m[:] = 1  # all ones: the leading run of zeros is empty
    r = rle_encode(m)
    assert r["size"] == list(m.shape) and sum(r["counts"]) == m.size and all(c > 0 for c in r["counts"][1:])
    assert np.array_equal(rle_decode(r), m)
    assert (r["counts"][0] == 0) == bool(m.flatten(order="F")[0])
    # COCO's definition, written out: runs over the Fortran-ordered pixels, alternating 0 / 1, starting with 0
    flat, runs, cur, n = m.flatten(order="F"), [], 0, 0
    for v in flat:
        if v == cur:
            n += 1
        else:
            runs.append(n)
            cur, n = v, 1
    runs.append(n)
    assert runs == r["counts"]


def test_constructor_contract():
    with pytest.raises(ValueError, match="network"):
        MaskGenerator()
    with pytest.raises(NotImplementedError, match="mocov2"):
        MaskGenerator(feature_types=["mocov2", "swav", "dino"], network=object())
    with pytest.raises(AssertionError):
        MaskGenerator(cluster_type="agglomerative", network=object())
    g = MaskGenerator(network=object(), cluster_type="k-means")
    assert g.cluster_sizes == (2, 3, 4) and g.feature_types == ["dino"]
