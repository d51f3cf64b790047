"""Where the step-level tolerances of the GPU tests come from: a rounding model, not GPU measurements.

The HIP path stores activations / conv operands in bf16 and accumulates in fp32.  oracle/model.py
can round at exactly those points -- and the activation gradients at the same points on the way back
(`emulate_bf16='grad'`); everything else of the oracle step stays fp32.  For every
step-level fixture of the GPU suite this script runs the CPU oracle step twice -- plain fp32 and bf16-emulating -- and
records N = |bf16-emulating - fp32| for each compared quantity: the change that bf16 storage ALONE makes to the
reference's numbers on that fixture, on the CPU.

Rule (tests/test_ssl_step_gpu.py: `tol()`): a correct bf16 implementation is another realisation of the same rounding
noise (it rounds at the same places, but to different neighbours: other summation orders inside the fp32 accumulators,
bf16 statistics of rounded instead of unrounded values, a re-associated head convolution), so against the fp32 oracle
it may deviate by about N, and by up to sqrt(2) N from the emulation.  Tolerance = 3 N, with a floor of 1e-3 relative
for scalars -- three "rounding-noise units".  One derived bound: the LENGTH of the gradient, a vector whose direction
carries rounding noise of angle theta (cos theta = N_cos, 0.90 on ResNet-101: the net is chaotic), is uncertain to
(1 - cos theta) / 2 -- a rotation by theta changes a projection by 1 - cos theta -- so the gradient-norm tolerance is
max(3 N_norm, (1 - N_cos) / 2): N_norm alone is one draw of a quantity that small.  No number measured on the GPU enters.

Writes tests/golden/bf16_tolerances.json.  Run in the build container:  python tests/golden/derive_tolerances.py
(CPU only; needs nothing from /root/reference: the fixtures are committed)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import labelpath as olp  # noqa: E402
from oracle import model as omodel  # noqa: E402
from oracle.step import CpuStep  # noqa: E402
from regda_amd.synthetic import make_batch  # noqa: E402


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def cosine(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


def noise(ref, emu, grad_names):
    """N per quantity between two CpuStep.step results (ref: fp32, emu: bf16-emulating)."""
    out = dict(loss_source=rel(emu['loss_source'], ref['loss_source']), loss_target=rel(emu['loss_target'], ref['loss_target']),
               loss_target_abs=abs(emu['loss_target'] - ref['loss_target']),
               grad_norm=rel(emu['grad_norm'], ref['grad_norm']),
               hard_mismatch=float((emu['hard'] != ref['hard']).float().mean()),
               soft_mean_abs=float((emu['soft'] - ref['soft']).abs().mean()))
    keep = [k for k in ref['grads'] if 'ppm.0.' not in k]       # the degenerate scale-1 branch is rounding noise in the reference itself
    out['grad_cos_global'] = cosine(torch.cat([emu['grads'][k].flatten() for k in keep]), torch.cat([ref['grads'][k].flatten() for k in keep]))
    out['grad_cos'] = {k: cosine(emu['grads'][k], ref['grads'][k]) for k in grad_names}
    out['grad_norm_ratio'] = {k: float(emu['grads'][k].norm() / (ref['grads'][k].norm() + 1e-30)) for k in grad_names}
    return out


def shallow_fixture(balancers):
    """tests/test_ssl_step_gpu.py::test_fused_step_matches_oracle_step / ..._with_class_balancing...: resnet17t, seed 6,
    batch seed 11, 4 + 4 images of 128 x 128, all-ones dropout masks, lr 1e-3."""
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=6)
    b = make_batch(b=4, size=128, seed=11, device='cpu')
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(1))
    ones = torch.ones(4, 512)
    res, steps = [], (2 if balancers else 1)
    for emu in (False, True):
        kw = {}
        if balancers:
            cb_s, cb_t = olp.ClassBalanceState(6, -1, 0.5, 0.5), olp.ClassBalanceState(6, -1, 0.5, 0.5)
            cb_s.freq = torch.tensor([0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
            cb_t.freq = torch.tensor([0.05, 0.05, 0.1, 0.1, 0.2, 0.5])
            kw = dict(balancer_s=cb_s, balancer_t=cb_t)
        cpu = CpuStep(sd, protos, resnet_type=rt, lr=1e-3, emulate_bf16=('grad' if emu else False), **kw)
        outs = [cpu.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], (ones, ones), (ones, ones))
                for _ in range(steps)]
        res.append((outs, cpu))
    (ref, cref), (emu, cemu) = res
    names = ['encoder.resnet.conv1.weight']
    n = [noise(r, e, names) for r, e in zip(ref, emu)]
    worst = {k: (min if 'cos' in k else max)(x[k] for x in n) for k in n[0] if not isinstance(n[0][k], dict)}
    worst['grad_cos'] = {k: min(x['grad_cos'][k] for x in n) for k in names}
    worst['grad_norm_ratio_dev'] = {k: max(abs(x['grad_norm_ratio'][k] - 1) for x in n) for k in names}
    worst['protos_rel'] = float((cemu.prototypes - cref.prototypes).norm() / cref.prototypes.norm())
    k = 'encoder.resnet.bn1.running_mean'
    worst['bn1_running_mean_abs'] = float((cemu.sd[k] - cref.sd[k]).abs().max())
    d_ref, d_emu = cref.sd[names[0]].detach() - sd[names[0]], cemu.sd[names[0]].detach() - sd[names[0]]
    worst['stem_update_cos'] = cosine(d_emu, d_ref)
    worst['stem_update_norm_dev'] = abs(float(d_emu.norm() / d_ref.norm()) - 1)
    if balancers:
        worst['freq_t_abs'] = float((cemu.balancer_t.freq - cref.balancer_t.freq).abs().max())
    return worst


def resnet101_fixture():
    """tests/test_ssl_step_gpu.py::test_resnet101_step_vs_reference_minted_step: the inputs of model_small.npz."""
    g = np.load(os.path.join(HERE, 'model_small.npz'))
    sd = omodel.init_state_dict('resnet101', 6, seed=1)
    xs, xt = torch.from_numpy(g['xs']), torch.from_numpy(g['xt'])
    lab = torch.from_numpy(g['lab_s'].astype(np.int64))
    soft_t = torch.from_numpy(g['soft_t'])
    regs = torch.from_numpy(g['regs'].astype(np.int64))
    ms = (torch.from_numpy(g['m5'][0]), torch.from_numpy(g['m6'][0]))
    mt = (torch.from_numpy(g['m5'][1]), torch.from_numpy(g['m6'][1]))
    names = ['encoder.resnet.conv1.weight', 'encoder.resnet.bn1.weight', 'encoder.resnet.bn1.bias', 'layer5.conv_last.4.weight',
             'layer5.conv_last.4.bias', 'layer6.conv_last.1.weight', 'encoder.resnet.layer4.2.bn3.bias', 'layer6.ppm.3.2.bias',
             'encoder.resnet.layer3.10.conv2.weight', 'layer5.conv_last.0.weight']
    res = []
    for emu in (False, True):
        cpu = CpuStep(sd, torch.from_numpy(g['protos']), resnet_type='resnet101', lr=1e-2, emulate_bf16=('grad' if emu else False))
        res.append((cpu.step(xs, lab, xt, soft_t, regs, ms, mt), cpu))
    (ref, cref), (emu, cemu) = res
    # the fp32 oracle step reproduces the reference-minted numbers (the oracle is pinned)
    assert rel(ref['loss_source'], float(g['loss_s'])) < 1e-4 and rel(ref['grad_norm'], float(g['grad_norm'])) < 1e-3
    n = noise(ref, emu, names)
    out = {k: v for k, v in n.items() if not isinstance(v, dict)}
    out['grad_cos_min'] = min(n['grad_cos'].values())
    out['grad_cos'] = n['grad_cos']
    out['grad_norm_ratio_dev_max'] = max(abs(v - 1) for v in n['grad_norm_ratio'].values())
    out['protos_rel'] = float((cemu.prototypes - cref.prototypes).norm() / cref.prototypes.norm())
    return out


def online_inputs(which):
    """tests/test_ssl_step_gpu.py::test_online_teacher_steps_match_the_oracle_online_steps: two consecutive steps with the
    ONLINE EMA teacher (soft labels from the teacher's eval forward inside the step, shadow update behind the optimizer;
    oracle/step.py: CpuStep(ema_decay=)).  'shallow': resnet17t, 4 + 4 images of 128 x 128; 'resnet101': 2 + 2 of 128 x 128."""
    if which == 'shallow':
        rt, sd, b, nb = 'resnet17t', omodel.init_state_dict('resnet17t', 6, seed=6), make_batch(b=4, size=128, seed=11, device='cpu'), 4
    elif which == 'resnet101_512':
        # the production tile, and WARM BatchNorm statistics: the reference's own buffers after 20 train-mode forwards
        # (tests/golden/eval_warm.npz; same weights).  The 128 x 128 fixture starts from the (0, 1) initialisation, where the
        # teacher's eval forward saturates and its mask noise says little about a run that is under way
        sd = eval_warm_inputs()[0]
        for head in ('layer5', 'layer6'):       # (undo eval_warm's classifier gain: ONLINE_CLS_GAIN is applied below)
            sd[f'{head}.conv_last.4.weight'] = sd[f'{head}.conv_last.4.weight'] / float(np.load(os.path.join(HERE, 'eval_warm.npz'))['cls_gain'])
        rt, b, nb = 'resnet101', make_batch(b=2, size=512, seed=12, device='cpu'), 2
    else:
        rt, sd, b, nb = 'resnet101', omodel.init_state_dict('resnet101', 6, seed=3, res_gamma=0.02), make_batch(b=2, size=128, seed=12, device='cpu'), 2
    # classifier gain 0.25: about half of the target pixels pass the teacher's 0.6 cut-off and the losses are O(1) (at the
    # fixtures' gain 1 the logits are +-30, a borderline pixel flips on the last bf16 bit and the target loss is ~10)
    for head in ('layer5', 'layer6'):
        sd[f'{head}.conv_last.4.weight'] = sd[f'{head}.conv_last.4.weight'] * ONLINE_CLS_GAIN
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(1))
    return rt, sd, b, protos, torch.ones(nb, 512)


ONLINE_DECAY, ONLINE_LR, ONLINE_STEPS, ONLINE_CLS_GAIN = 0.9, 1e-3, 2, 0.25


def online_fixture(which):
    rt, sd, b, protos, ones = online_inputs(which)
    res = []
    for emu in (False, True):
        cpu = CpuStep(sd, protos, resnet_type=rt, lr=ONLINE_LR, emulate_bf16=('grad' if emu else False), ema_decay=ONLINE_DECAY)
        outs = []
        for _ in range(ONLINE_STEPS):
            o = cpu.step(b['images_s'], b['label_s'], b['images_t'], None, b['regs_t'], (ones, ones), (ones, ones))
            o['teacher_soft'] = cpu.last_soft_t
            outs.append(o)
        res.append((outs, cpu))
    (ref, cref), (emu, cemu) = res
    names = ['encoder.resnet.conv1.weight']
    n = [noise(r, e, names) for r, e in zip(ref, emu)]
    worst = {k: (min if 'cos' in k else max)(x[k] for x in n) for k in n[0] if not isinstance(n[0][k], dict)}
    worst['teacher_soft_mean_abs'] = max(float((e['teacher_soft'] - r['teacher_soft']).abs().mean()) for r, e in zip(ref, emu))
    worst['labelled_fraction'] = min(float((r['hard'] >= 0).float().mean()) for r in ref)
    worst['protos_rel'] = float((cemu.prototypes - cref.prototypes).norm() / cref.prototypes.norm())
    k = names[0]
    # the shadow after two updates, relative to how far it has moved from the initial weights
    mv = torch.cat([(cref.shadow[q] - sd[q]).flatten() for q in cref.names])
    dv = torch.cat([(cemu.shadow[q] - cref.shadow[q]).flatten() for q in cref.names])
    worst['shadow_move_rel'] = float(dv.norm() / mv.norm())
    return worst


TRAJ_STEPS = 20


def trajectory_inputs():
    """tests/test_ssl_step_gpu.py::test_twenty_step_loss_curve_tracks_the_oracle: shallow topology, two alternating
    batches of 4 + 4 images of 128 x 128, the reference's schedule shape (warm-up then constant) at a rate that moves the
    loss visibly in 20 steps."""
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=9)
    batches = [make_batch(b=4, size=128, seed=31, device='cpu'), make_batch(b=4, size=128, seed=32, device='cpu')]
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(4))
    ones = torch.ones(4, 512)
    lrs = [1e-3 * min(1.0, (i + 1) / 5.0) for i in range(TRAJ_STEPS)]
    return rt, sd, batches, protos, ones, lrs


def run_trajectory(emulate):
    rt, sd, batches, protos, ones, lrs = trajectory_inputs()
    cpu = CpuStep(sd, protos, resnet_type=rt, lr=1e-3, emulate_bf16=('grad' if emulate else False))
    out = []
    for i, lr in enumerate(lrs):
        b = batches[i % 2]
        r = cpu.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], (ones, ones), (ones, ones), lr=lr)
        out.append((r['loss_source'], r['loss_target'], r['grad_norm']))
    return out


def trajectory_fixture():
    """N per step of a 20-step run: the bf16-emulating oracle's loss curve against the fp32 oracle's."""
    ref, emu = run_trajectory(False), run_trajectory(True)
    ds = [e[0] - r[0] for r, e in zip(ref, emu)]
    dt = [e[1] - r[1] for r, e in zip(ref, emu)]
    return dict(loss_source_abs_max=float(np.max(np.abs(ds))), loss_target_abs_max=float(np.max(np.abs(dt))),
                loss_source_abs_mean_signed=float(np.mean(ds)), loss_target_abs_mean_signed=float(np.mean(dt)),
                ref_loss_source=[r[0] for r in ref], ref_loss_target=[r[1] for r in ref],
                emu_loss_source_abs=ds, emu_loss_target_abs=dt)


def resnet101_mid_fixture():
    """tests/test_ssl_step_gpu.py::test_resnet101_step_vs_reference_minted_step_128: the inputs of model_mid.npz."""
    g = np.load(os.path.join(HERE, 'model_mid.npz'))
    sd = omodel.init_state_dict('resnet101', 6, seed=3, res_gamma=0.02)
    t = lambda k: torch.from_numpy(g[k])
    ms = (t('m5')[0], t('m6')[0])
    mt = (t('m5')[1], t('m6')[1])
    keys = [k[5:] for k in g.files if k.startswith('grad:')]
    res = []
    for emu in (False, True):
        cpu = CpuStep(sd, t('protos'), resnet_type='resnet101', lr=1e-2, emulate_bf16=('grad' if emu else False))
        res.append((cpu.step(t('xs'), t('lab_s').long(), t('xt'), t('soft_t'), t('regs').long(), ms, mt), cpu))
    (ref, cref), (emu, cemu) = res
    assert rel(ref['loss_source'], float(g['loss_s'])) < 1e-4 and rel(ref['grad_norm'], float(g['grad_norm'])) < 5e-3
    n = noise(ref, emu, [])
    out = {k: v for k, v in n.items() if not isinstance(v, dict)}

    def sl(name, tensors):
        base, _, s_ = name.partition('[')
        return tensors[base][:int(s_[1:-1])] if s_ else tensors[base]
    out['grad_cos'] = {k: cosine(sl(k, emu['grads']), sl(k, ref['grads'])) for k in keys}
    out['grad_cos_min'] = min(out['grad_cos'].values())
    out['grad_norm_ratio_dev_max'] = max(abs(float(sl(k, emu['grads']).norm() / (sl(k, ref['grads']).norm() + 1e-30)) - 1) for k in keys)
    out['protos_rel'] = float((cemu.prototypes - cref.prototypes).norm() / cref.prototypes.norm())
    return out


FULL_GRAD_NAMES = ['encoder.resnet.conv1.weight', 'encoder.resnet.bn1.weight', 'encoder.resnet.layer1.0.conv1.weight',
                   'encoder.resnet.layer1.2.conv3.weight', 'encoder.resnet.layer2.1.conv2.weight',
                   'encoder.resnet.layer2.3.bn3.weight', 'encoder.resnet.layer3.0.downsample.0.weight',
                   'encoder.resnet.layer3.5.conv2.weight', 'encoder.resnet.layer3.11.bn2.weight',
                   'encoder.resnet.layer3.17.conv1.weight', 'encoder.resnet.layer3.22.conv3.weight',
                   'encoder.resnet.layer4.0.conv2.weight', 'encoder.resnet.layer4.2.conv3.weight', 'layer5.ppm.3.1.weight',
                   'layer5.conv_last.0.weight', 'layer6.conv_last.1.bias', 'layer6.conv_last.4.weight']


# Residual-branch gain of the full-size fixture.  What bf16 storage does to the per-layer gradient DIRECTIONS of a random-init
# ResNet-101 at 2 + 2 x 512 x 512 (this script, cosine between the bf16-emulating and the fp32 oracle, 17 tensors over
# depth): res_gamma 0.1 (the other fixtures) 0.877 - 0.967 in the backbone, 0.02 0.959 - 0.985, 0.004 0.963 - 0.988 -- the
# noise floor of a 101-layer BatchNorm net in bf16 is ~0.97, however small the residual gain; 0.02 is used.
FULL_RES_GAMMA = 0.02


def full_size_inputs(res_gamma=FULL_RES_GAMMA):
    """The inputs of tests/test_ssl_step_gpu.py::test_full_size_resnet101_step_vs_oracle (BASELINE config[0]'s shape:
    ResNet-101, 2 + 2 images of 512 x 512, offline soft labels): shared by the test and by this derivation."""
    sd = omodel.init_state_dict('resnet101', 6, seed=5, res_gamma=res_gamma)
    b = make_batch(b=2, size=512, seed=77, device='cpu')
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(2))
    ones = torch.ones(2, 512)
    return sd, b, protos, ones


def resnet101_full_fixture(res_gamma=FULL_RES_GAMMA):
    sd, b, protos, ones = full_size_inputs(res_gamma)
    res = []
    for emu in (False, True):
        cpu = CpuStep(sd, protos, resnet_type='resnet101', lr=1e-3, emulate_bf16=('grad' if emu else False))
        res.append((cpu.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], (ones, ones), (ones, ones)), cpu))
    (ref, cref), (emu, cemu) = res
    n = noise(ref, emu, FULL_GRAD_NAMES)
    out = {k: v for k, v in n.items() if not isinstance(v, dict)}
    out['grad_cos_min'] = min(n['grad_cos'].values())
    out['grad_cos'] = n['grad_cos']
    out['grad_norm_ratio_dev_max'] = max(abs(v - 1) for v in n['grad_norm_ratio'].values())
    out['protos_rel'] = float((cemu.prototypes - cref.prototypes).norm() / cref.prototypes.norm())
    out['labelled_fraction'] = float((ref['hard'] >= 0).float().mean())
    return out


def config1_inputs(res_gamma=FULL_RES_GAMMA):
    """The inputs of tests/test_ssl_step_gpu.py::test_full_size_config1_step_vs_oracle: BASELINE config[1]'s batch --
    ResNet-101, 8 + 8 images of 512 x 512 (offline soft labels, all-ones dropout masks so both sides see the same net)."""
    sd = omodel.init_state_dict('resnet101', 6, seed=5, res_gamma=res_gamma)
    b = make_batch(b=8, size=512, seed=78, device='cpu')
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(2))
    ones = torch.ones(8, 512)
    return sd, b, protos, ones


def running_stat_noise(sd_a, sd_b):
    """Distance of the BatchNorm running statistics of two state dicts, max over the layers: running_mean in units of the
    layer's standard deviation, ||mean_a - mean_b|| / ||sqrt(var_b)|| (a relative measure would divide by ~0 wherever the
    true mean is 0 -- the scale-1 PPM branch's input is an instance-normalised global average); running_var relative."""
    out = {'running_mean': 0.0, 'running_var': 0.0}
    for k in sd_b:
        if k.endswith('running_mean'):
            sv = sd_b[k[:-len('running_mean')] + 'running_var'].double().clamp_min(0).sqrt()
            out['running_mean'] = max(out['running_mean'], float((sd_a[k].double() - sd_b[k].double()).norm() / (sv.norm() + 1e-30)))
        elif k.endswith('running_var'):
            out['running_var'] = max(out['running_var'], float((sd_a[k].double() - sd_b[k].double()).norm() / (sd_b[k].double().norm() + 1e-30)))
    return out


def resnet101_config1_fixture(res_gamma=FULL_RES_GAMMA):
    """~4 minutes on 8 cores, ~16 GB: the fp32 and the bf16-emulating oracle step at 8 + 8 x 512 x 512."""
    sd, b, protos, ones = config1_inputs(res_gamma)
    res = []
    for emu in (False, True):
        cpu = CpuStep(sd, protos, resnet_type='resnet101', lr=1e-3, emulate_bf16=('grad' if emu else False))
        res.append((cpu.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], (ones, ones), (ones, ones)), cpu))
    (ref, cref), (emu, cemu) = res
    n = noise(ref, emu, FULL_GRAD_NAMES)
    out = {k: v for k, v in n.items() if not isinstance(v, dict)}
    out['grad_cos_min'] = min(n['grad_cos'].values())
    out['grad_cos'] = n['grad_cos']
    out['grad_norm_ratio_dev_max'] = max(abs(v - 1) for v in n['grad_norm_ratio'].values())
    out['protos_rel'] = float((cemu.prototypes - cref.prototypes).norm() / cref.prototypes.norm())
    out['labelled_fraction'] = float((ref['hard'] >= 0).float().mean())
    rs = running_stat_noise({k: v.detach() for k, v in cemu.sd.items()}, {k: v.detach() for k, v in cref.sd.items()})
    out['bn_running_mean_rel'], out['bn_running_var_rel'] = rs['running_mean'], rs['running_var']
    return out


def eval_warm_inputs():
    """tests/golden/eval_warm.npz (minted from the reference by make_goldens.gold_eval_warm): the state dict with the
    reference's warm BatchNorm buffers, the eval input, the reference's probabilities."""
    g = np.load(os.path.join(HERE, 'eval_warm.npz'))
    sd = omodel.init_state_dict('resnet101', 6, seed=3, res_gamma=0.02)
    for head in ('layer5', 'layer6'):
        sd[f'{head}.conv_last.4.weight'] = sd[f'{head}.conv_last.4.weight'] * float(g['cls_gain'])
    off = 0
    for k, n in zip(g['buffer_names'], g['buffer_sizes']):
        sd[str(k)] = torch.from_numpy(g['buffers'][off:off + int(n)].copy())
        off += int(n)
    for k in sd:
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(int(g['nbt']))
    return sd, torch.from_numpy(g['xe']), torch.from_numpy(g['probs'])


EVAL_DECISIVE_MARGIN = 0.05


def eval_agreement(got, ref):
    """mean |p - p_ref|, argmax agreement over all pixels, and over the pixels where the reference is DECISIVE (top-1 minus
    top-2 probability > EVAL_DECISIVE_MARGIN) together with their share."""
    top2 = ref.topk(2, 1)[0]
    dec = (top2[:, 0] - top2[:, 1]) > EVAL_DECISIVE_MARGIN
    same = got.argmax(1) == ref.argmax(1)
    return dict(mean_abs=float((got - ref).abs().mean()), argmax_agree=float(same.float().mean()),
                decisive_share=float(dec.float().mean()), decisive_disagree=float((~same & dec).float().sum() / dec.float().sum()))


def eval_warm_fixture():
    sd, xe, ref = eval_warm_inputs()
    with torch.no_grad():
        a = omodel.forward(sd, xe, False, None, 'resnet101')
        b = omodel.forward(sd, xe, False, None, 'resnet101', emulate_bf16=True)
    assert float((a - ref).abs().max()) < 1e-5         # the oracle IS the reference here (pinned by tests/test_oracle_golden.py)
    out = eval_agreement(b, a)
    out['argmax_disagree'] = 1.0 - out['argmax_agree']
    return out


def model_fixture(rt, sd, xs, lab, masks):
    """tests/test_model_gpu.py::_run_case: one train-mode forward + loss + backward of the network alone."""
    names = omodel.param_names(sd)
    keep = [k for k in names if 'ppm.0.' not in k]          # degenerate branch, see oracle.model.init_state_dict
    res = []
    for emu in (False, 'grad'):
        sdr = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
        r1, r2, rf = omodel.forward(sdr, xs, True, masks, rt, {}, None, emulate_bf16=emu)
        loss = olp.loss_calc([r1, r2], lab, -1)
        g = torch.autograd.grad(loss, [sdr[k] for k in names])
        flat = torch.cat([t.reshape(-1) for k, t in zip(names, g) if k in keep])
        res.append((r1.detach(), r2.detach(), rf.detach(), float(loss), flat))
    (a1, a2, af, al, ag), (b1, b2, bf, bl, bg) = res
    l2 = lambda x, y: float((x - y).norm() / (y.norm() + 1e-12))
    return dict(x1=l2(b1, a1), x2=l2(b2, a2), feat=l2(bf, af), loss=rel(bl, al), grad_cos=cosine(bg, ag),
                grad_norm=abs(float(bg.norm() / ag.norm()) - 1))


def shallow_model_fixture():
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=1)
    gen = torch.Generator().manual_seed(3)
    xs = torch.randn(4, 3, 128, 128, generator=gen)
    masks = ((torch.rand(4, 512, generator=gen) > 0.1).to(torch.uint8), (torch.rand(4, 512, generator=gen) > 0.1).to(torch.uint8))
    lab = torch.from_numpy(np.kron(np.random.default_rng(0).integers(-1, 6, size=(4, 8, 8)), np.ones((16, 16), np.int64)))
    return model_fixture(rt, sd, xs, lab, masks)


def resnet101_model_fixture():
    g = np.load(os.path.join(HERE, 'model_small.npz'))
    sd = omodel.init_state_dict('resnet101', 6, seed=1)
    masks = (torch.from_numpy(g['m5'][0]), torch.from_numpy(g['m6'][0]))
    return model_fixture('resnet101', sd, torch.from_numpy(g['xs']), torch.from_numpy(g['lab_s'].astype(np.int64)), masks)


if __name__ == '__main__':
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    if len(sys.argv) > 2 and sys.argv[1] == '--only':       # recompute the named fixtures, keep the rest of the table
        path = os.path.join(HERE, 'bf16_tolerances.json')
        table = json.load(open(path))
        fns = {'shallow_online': lambda: online_fixture('shallow'), 'resnet101_online_128': lambda: online_fixture('resnet101'),
               'resnet101_config1': lambda: dict(resnet101_config1_fixture(FULL_RES_GAMMA), res_gamma=FULL_RES_GAMMA),
               'resnet101_online_512': lambda: online_fixture('resnet101_512'),
               'resnet101_eval_warm': eval_warm_fixture}
        for name in sys.argv[2:]:
            table[name] = fns[name]()
            print(name, json.dumps(table[name], indent=1, sort_keys=True))
        with open(path, 'w') as f:
            json.dump(table, f, indent=1, sort_keys=True)
        sys.exit(0)
    out = {'rule': 'tolerance = max(3 * N, floor); N = |bf16-emulating oracle - fp32 oracle| on the fixture (CPU); '
                   'cosines: 1 - tol_cos = 3 * (1 - N_cos)',
           'factor': 3.0,
           'shallow_step': shallow_fixture(False),
           'shallow_step_class_balancing': shallow_fixture(True),
           'resnet101_step': resnet101_fixture(),
           'resnet101_step_mid': resnet101_mid_fixture(),
           'shallow_trajectory': trajectory_fixture(),
           'resnet101_full': dict(resnet101_full_fixture(FULL_RES_GAMMA), res_gamma=FULL_RES_GAMMA),
           'shallow_online': online_fixture('shallow'),
           'resnet101_online_128': online_fixture('resnet101'),
           'resnet101_config1': dict(resnet101_config1_fixture(FULL_RES_GAMMA), res_gamma=FULL_RES_GAMMA),
           'resnet101_online_512': online_fixture('resnet101_512'),
           'resnet101_eval_warm': eval_warm_fixture(),
           'shallow_model': shallow_model_fixture(),
           'resnet101_model': resnet101_model_fixture()}
    with open(os.path.join(HERE, 'bf16_tolerances.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
