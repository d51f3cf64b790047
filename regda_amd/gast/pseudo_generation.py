"""pseudo_selection -- mirror of regda/gast/pseudo_generation.py:59-93."""
from .. import ops


def pseudo_selection(mask, cutoff_top=0.8, cutoff_low=0.6, return_type='ndarray', ignore_label=-1):
    """soft (b,c,h,w) probabilities -> hard labels (b,h,w) int64; same arguments, asserts and
    return types as the reference."""
    assert return_type in ['ndarray', 'tensor']
    ret = ops.pseudo_select(mask, cutoff_top, cutoff_low, ignore_label, check=True)
    if return_type == 'ndarray':
        return ret.cpu().numpy()
    return ret
