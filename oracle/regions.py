"""ORACLE (test infrastructure only -- see oracle/__init__.py): the reference's assembly of a SAM region map,
regda/utils/local_region_homog.py:51-56 inside SAM.get_local_regions: the masks of the automatic mask generator in
generator order, those with `area >= area_thrshold` painted one over the other as ids i + 1, background 0, int32.
The generator itself (third-party segment_anything ViT-H) is not restated.  Pinned by tests/golden/regions.npz, minted
from the reference's own loop fed with synthetic annotations (tests/golden/make_goldens.py: gold_regions)."""
import numpy as np


def regions_from_masks(masks, areas, area_threshold=1024):
    """masks (K, H, W) bool / uint8, areas (K,) -> (H, W) int32."""
    masks = np.asarray(masks).astype(bool)
    out = np.zeros(masks.shape[1:], np.float64)                 # the reference paints into np.zeros(size): float64
    for i in range(masks.shape[0]):
        if areas[i] >= area_threshold:                          # :53
            out[masks[i]] = i + 1                               # :55 later masks overwrite earlier ones
    return out.astype(np.int32)                                 # :56
