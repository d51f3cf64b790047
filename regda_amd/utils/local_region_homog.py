"""Local Region Homogenizing (LRH) -- mirror of regda/utils/local_region_homog.py:99-152.

Same constructor and call signature as the reference `Homogenizer`; the work is
done by the HIP kernels behind `rgda_lrh` (bit-exact against the reference,
tests/test_label_gpu.py).

`SAM` mirrors the reference wrapper (:27-64) as far as the reference's own code goes: the assembly of the int32 region
map from the automatic mask generator's output (`rgda_masks_to_regions`, bit-exact).  The generator itself -- third-party
segment_anything, ViT-H -- is NOT built here: pass any object with `.generate(image) -> [{'segmentation', 'area'}]`
(a `SamAutomaticMaskGenerator` configured as local_region_homog.py:32-39).  Reading the image file and writing the
`.tif` (cv2 / skimage) stay with the caller.
"""
import numpy as np
import torch

from .. import ops


def regions_from_anns(anns, size, area_thrshold=1024, device='cuda'):
    """The region map of one image from the mask generator's annotations (local_region_homog.py:51-56): masks with
    `area >= area_thrshold`, painted in generator order, ids i + 1, background 0 -> (H, W) int32 tensor on the GPU."""
    H, W = size
    if len(anns) == 0:
        return torch.zeros((H, W), dtype=torch.int32, device=device)
    masks = torch.from_numpy(np.stack([np.asarray(a['segmentation']) for a in anns]).astype(np.uint8)).to(device)
    areas = torch.tensor([int(a['area']) for a in anns], dtype=torch.int64, device=device)
    assert masks.shape[1:] == (H, W)
    return ops.masks_to_regions(masks, areas, area_thrshold)


class SAM:
    def __init__(self, mask_generator):
        """mask_generator: the third-party `SamAutomaticMaskGenerator` (points_per_side=32, pred_iou_thresh=0.90,
        stability_score_thresh=0.95, crop_n_layers=1, crop_n_points_downscale_factor=2, local_region_homog.py:32-39)
        or anything with the same `.generate(image)`."""
        self.model = mask_generator

    def get_local_regions(self, image, area_thrshold=1024):
        """image: (H, W, 3) uint8 RGB array (the caller reads the file) -> (H, W) int32 numpy region map, what the
        reference saves as `reg_dir/<name>.tif`."""
        image = np.asarray(image)
        anns = self.model.generate(image)
        return regions_from_anns(anns, image.shape[:2], area_thrshold).cpu().numpy()


class Homogenizer(torch.nn.Module):
    def __init__(self, percent=0.9, class_num=6, ignore_label=255, max_regions=4096, check=True):
        """`max_regions` (extension): exclusive upper bound on region ids; the reference sizes its
        histogram by `regions.max()+1` with a host sync (torch_scatter), we take a static bound and
        keep the call sync-free when `check=False` (the flag word is then checked by the caller)."""
        super().__init__()
        self.percent = percent
        self.class_num = class_num
        self.ignore_label = ignore_label
        self.max_regions = max_regions
        self.check = check

    def forward(self, pseudo_labels, regions):
        assert pseudo_labels.dim() == 3                       # local_region_homog.py:133
        return ops.lrh(pseudo_labels, regions, self.percent, self.class_num, self.ignore_label,
                       self.max_regions, self.check)
