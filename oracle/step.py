"""One full SSL iteration on CPU fp32 (stock PyTorch autograd).  TEST INFRASTRUCTURE ONLY.

Restates tools/train_ssl_reg.py:198-241: two train-mode forwards, label_refine,
pseudo_selection, LRH, update_prototype, 2x loss_calc, backward,
clip_grad_norm_(32), SGD(momentum .9, wd 5e-4).  Used by tests as the end-to-end
checker and by bench.py's `cpu_baseline` leg (kind "port").
"""
import torch

from . import labels, labelpath, model


class CpuStep:
    def __init__(self, sd, prototypes, resnet_type='resnet101', class_num=6, ignore_label=-1,
                 lr=1e-2, momentum=0.9, weight_decay=5e-4, max_norm=32.0,
                 cutoff_top=0.8, cutoff_low=0.6, percent=0.5, proto_decay=0.996,
                 refine_temp=2.0, sam_refine=True, balancer_s=None, balancer_t=None, emulate_bf16=False, ema_decay=None):
        # balancer_s / balancer_t: labelpath.ClassBalanceState (--bcs / --bct, train_ssl_reg.py:125-158) or None
        self.balancer_s, self.balancer_t = balancer_s, balancer_t
        # emulate_bf16: the network rounds to bf16 where the HIP path stores bf16 (oracle/model.py: forward); fp32
        # accumulation, fp32 label path and optimizer.  tests/golden/derive_tolerances.py uses the difference between the
        # two modes as the rounding-noise scale the GPU tolerances are stated in.
        self.emulate_bf16 = emulate_bf16
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = model.param_names(self.sd)
        for k in self.names:
            self.sd[k].requires_grad_(True)
        self.mom = {k: None for k in self.names}
        self.prototypes = prototypes.clone()
        self.rt = resnet_type
        self.C, self.ig = class_num, ignore_label
        self.lr, self.m, self.wd, self.max_norm = lr, momentum, weight_decay, max_norm
        self.top, self.low, self.percent = cutoff_top, cutoff_low, percent
        self.pdecay, self.temp, self.sam = proto_decay, refine_temp, sam_refine
        # ema_decay: the ONLINE EMA teacher (BASELINE.json north_star; regda/utils/ema.py:34-51).  register(): the shadow
        # starts as a copy of the parameters (:44-47); `step(soft_t=None)` takes the target soft labels from the teacher's
        # eval forward (Encoder.py:152-155: (softmax(up x1) + softmax(up x2)) / 2) on the shadow weights with the
        # student's BatchNorm buffers as they stand at the START of the step (ema.py averages parameters only), and
        # update() (:49-54) follows the optimizer: shadow = (1 - d) * param + d * shadow
        self.ema_decay = ema_decay
        self.shadow = None if ema_decay is None else {k: self.sd[k].detach().clone() for k in self.names}

    def step(self, images_s, label_s, images_t, soft_t, regs_t, drop_masks_s=None, drop_masks_t=None,
             lr=None, phases=None):
        import time
        t0 = time.time()
        sd = self.sd
        if soft_t is None:
            assert self.shadow is not None, 'soft_t=None needs the online teacher (ema_decay=...)'
            with torch.no_grad():
                tsd = {k: (self.shadow[k] if k in self.shadow else v.detach()) for k, v in sd.items()}
                soft_t = model.forward(tsd, images_t, False, None, self.rt, emulate_bf16=self.emulate_bf16)
            self.last_soft_t = soft_t
        t_teacher = time.time()
        ns = {}
        s1, s2, feat_s = model.forward(sd, images_s, True, drop_masks_s, self.rt, ns, emulate_bf16=self.emulate_bf16)
        for k, v in ns.items():
            sd[k] = v
        ns = {}
        t1, t2, feat_t = model.forward(sd, images_t, True, drop_masks_t, self.rt, ns, emulate_bf16=self.emulate_bf16)
        for k, v in ns.items():
            sd[k] = v
        t_fwd = time.time()
        with torch.no_grad():
            soft = labelpath.label_refine(feat_t, self.prototypes, [t1, t2], soft_t, True, 'all', self.temp)
            hard = torch.from_numpy(labels.pseudo_selection(soft.numpy(), self.top, self.low, self.ig))
            if self.sam:
                hard = torch.from_numpy(labels.homogenize(hard.numpy(), regs_t.squeeze(1).numpy(),
                                                          self.percent, self.C, self.ig))
            self.prototypes, _ = labelpath.update_prototype(feat_s, label_s, self.prototypes,
                                                            self.pdecay, self.C, self.ig)
        loss_s = labelpath.loss_calc([s1, s2], label_s, self.ig, self.balancer_s)
        loss_t = labelpath.loss_calc([t1, t2], hard, self.ig, self.balancer_t)
        loss = loss_s + loss_t
        t_lab = time.time()
        params = [sd[k] for k in self.names]
        grads = torch.autograd.grad(loss, params)
        t_bwd = time.time()
        with torch.no_grad():
            total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
            coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)   # clip_grad_norm_
            lr = self.lr if lr is None else lr
            for k, g in zip(self.names, grads):
                g = g * coef
                p = sd[k]
                d = g + self.wd * p
                if self.mom[k] is None:
                    self.mom[k] = d.clone()
                else:
                    self.mom[k].mul_(self.m).add_(d)
                p.sub_(lr * self.mom[k])
                if self.shadow is not None:                                     # ema.update(), ema.py:49-54
                    self.shadow[k] = (1.0 - self.ema_decay) * p.detach() + self.ema_decay * self.shadow[k]
        t_opt = time.time()
        if phases is not None:
            phases.update(teacher=t_teacher - t0, fwd=t_fwd - t_teacher, label=t_lab - t_fwd, bwd=t_bwd - t_lab, opt=t_opt - t_bwd)
        return dict(loss=float(loss.detach()), loss_source=float(loss_s.detach()), loss_target=float(loss_t.detach()),
                    grad_norm=float(total), hard=hard, soft=soft, grads=dict(zip(self.names, grads)),
                    preds=(s1.detach(), s2.detach(), t1.detach(), t2.detach()),
                    feats=(feat_s.detach(), feat_t.detach()))


class CpuAlignStep(CpuStep):
    """One stage-2 iteration, tools/train_align_reg.py:144-196 (defaults: --align-domain 0, --refine-label 1,
    --refine-mode all, --pcl-temp 8): source forward, update_prototype, target forward, the student's own
    (softmax(up x1) + softmax(up x2)) / 2 as soft labels, label_refine, pseudo_selection, LRH, DownscaleLabel,
    loss = CE(source) + 0.5 * (PCL(source) + PCL(target)), backward, clip, SGD."""

    def __init__(self, *a, pcl_temp=8.0, **k):
        super().__init__(*a, **k)
        self.pcl_temp = pcl_temp

    def step(self, images_s, label_s, images_t, regs_t, drop_masks_s=None, drop_masks_t=None, lr=None):
        import torch.nn.functional as F
        sd = self.sd
        ns = {}
        s1, s2, feat_s = model.forward(sd, images_s, True, drop_masks_s, self.rt, ns)
        for k, v in ns.items():
            sd[k] = v
        with torch.no_grad():
            self.prototypes, label_s_down = labelpath.update_prototype(feat_s, label_s, self.prototypes, self.pdecay,
                                                                       self.C, self.ig)
        ns = {}
        t1, t2, feat_t = model.forward(sd, images_t, True, drop_masks_t, self.rt, ns)
        for k, v in ns.items():
            sd[k] = v
        with torch.no_grad():
            size = images_t.shape[-2:]
            x1 = F.interpolate(t1, size, mode='bilinear', align_corners=True)
            x2 = F.interpolate(t2, size, mode='bilinear', align_corners=True)
            soft_t = (x1.softmax(dim=1) + x2.softmax(dim=1)) * 0.5
            soft = labelpath.label_refine(feat_t, self.prototypes, [t1, t2], soft_t, True, 'all', self.temp)
            hard = torch.from_numpy(labels.pseudo_selection(soft.numpy(), self.top, self.low, self.ig))
            if self.sam:
                hard = torch.from_numpy(labels.homogenize(hard.numpy(), regs_t.squeeze(1).numpy(),
                                                          self.percent, self.C, self.ig))
            label_t = torch.from_numpy(labels.downscale_label(hard.numpy(), 16, self.C, self.ig, 0.75))
        loss_seg = labelpath.loss_calc([s1, s2], label_s, self.ig)
        loss_align = (labelpath.prototype_contrastive_loss(self.prototypes, feat_s, label_s_down, self.pcl_temp, self.ig) +
                      labelpath.prototype_contrastive_loss(self.prototypes, feat_t, label_t, self.pcl_temp, self.ig)) * 0.5
        loss = loss_seg + loss_align
        params = [sd[k] for k in self.names]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)]
        with torch.no_grad():
            total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
            coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
            lr = self.lr if lr is None else lr
            for k, g in zip(self.names, grads):
                g = g * coef
                p = sd[k]
                d = g + self.wd * p
                if self.mom[k] is None:
                    self.mom[k] = d.clone()
                else:
                    self.mom[k].mul_(self.m).add_(d)
                p.sub_(lr * self.mom[k])
        return dict(loss=float(loss.detach()), loss_seg=float(loss_seg.detach()), loss_align=float(loss_align.detach()),
                    grad_norm=float(total), hard=hard, label_t=label_t, label_s_down=label_s_down,
                    grads=dict(zip(self.names, grads)), feats=(feat_s.detach(), feat_t.detach()))
