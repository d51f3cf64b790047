"""Local Region Homogenizing (LRH) -- mirror of regda/utils/local_region_homog.py:99-152.

Same constructor and call signature as the reference `Homogenizer`; the work is
done by the HIP kernels behind `rgda_lrh` (bit-exact against the reference,
tests/test_label_gpu.py).  The SAM wrapper / get_all_regs of the reference file
(offline region-map generation, third-party segment_anything) are out of scope.
"""
import torch

from .. import ops


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
