"""Where the online teacher's pseudo-label noise comes from -- answered with the CPU rounding
model, no GPU: the EMA teacher's eval forward (oracle/model.py, Encoder.py:152-155) on the `resnet101_online_*` fixtures'
inputs in fp32, with bf16 storage everywhere (the model of the HIP path), and with bf16 storage at ONE group of storage
points at a time (`emulate_where`).  Reported per variant: mean |soft - soft_fp32| and the fraction of pixels whose
pseudo_selection label (0.8 / 0.6 cut-offs) differs from the fp32 teacher's.

    python tests/golden/attribute_teacher_noise.py [128] [512]      -> tests/golden/teacher_noise_attribution.json

Reading of the committed table: DESIGN.md section 5 (round 6)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import labels as olab  # noqa: E402
from oracle import model as omodel  # noqa: E402
import derive_tolerances as D  # noqa: E402

GROUPS = ['weights', 'stem', 'layer1', 'layer2', 'layer3', 'layer4', 'head', 'tail']


def run(size):
    which = 'resnet101' if size == 128 else 'resnet101_512'
    rt, sd, b, protos, ones = D.online_inputs(which)
    x = b['images_t']
    with torch.no_grad():
        ref = omodel.forward(sd, x, False, None, rt)
        hard_ref = olab.pseudo_selection(ref.numpy(), 0.8, 0.6, -1)
        rows = {}
        variants = [('all', None)] + [(g, {g}) for g in GROUPS] + [('all but tail', set(GROUPS) - {'tail'}),
                                                                   ('all but layer3', set(GROUPS) - {'layer3'})]
        for name, where in variants:
            soft = omodel.forward(sd, x, False, None, rt, emulate_bf16=True, emulate_where=where)
            hard = olab.pseudo_selection(soft.numpy(), 0.8, 0.6, -1)
            rows[name] = dict(soft_mean_abs=float((soft - ref).abs().mean()), hard_mismatch=float((hard != hard_ref).mean()))
            print('%4d  %-16s soft %.3e  hard flips %.4f' % (size, name, rows[name]['soft_mean_abs'], rows[name]['hard_mismatch']), flush=True)
    return dict(labelled_fraction=float((hard_ref >= 0).mean()), variants=rows)


if __name__ == '__main__':
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sizes = [int(a) for a in sys.argv[1:]] or [128, 512]
    path = os.path.join(HERE, 'teacher_noise_attribution.json')
    out = json.load(open(path)) if os.path.exists(path) else {}
    for s in sizes:
        out[str(s)] = run(s)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
