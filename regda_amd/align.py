"""The stage-2 ("align") inner loop of tools/train_align_reg.py:144-196 as one fused, sync-free step (SURVEY.md 8f
rank 2), on the same kernel plans as the SSL step (regda_amd/ssl.py):

    model(src) -> update_prototype -> model(tgt) -> the student's own (softmax(up x1) + softmax(up x2)) / 2 as soft
    labels -> label_refine -> pseudo_selection -> Homogenizer (LRH) -> DownscaleLabel
    loss = loss_calc(src) + 0.5 * (PrototypeContrastiveLoss(src) + PrototypeContrastiveLoss(tgt))     [--align-domain 0]
    -> backward -> clip_grad_norm_(32) -> SGD

Differences to the SSL step that matter for the kernels: there is no CE on the target logits (their gradient is
zero) and the loss reaches the network through the third forward output, the instance-normalised features
(rgda_pcl_loss writes d loss / d feat pixel-major, `Deeplabv2._backward_plan(gfeat=...)` adds it in the
instance-norm backward)."""
import torch

from . import ops
from .ddp import all_reduce_prototype_statistics
from .ssl import SSLStep

BF = torch.bfloat16


class AlignStep(SSLStep):
    def __init__(self, model, prototypes, pcl_temperature=8.0, **kw):
        kw.setdefault('proto_decay', 0.999)        # Aligner(decay=0.999), train_align_reg.py:112-113
        kw['ema_decay'] = None
        super().__init__(model, prototypes, **kw)
        self.pcl_temp = pcl_temperature
        self.loss_align = torch.zeros(1, device=model.device)

    def step(self, images_s, label_s, images_t, regs_t, lr):
        """One stage-2 iteration.  Returns device tensors (loss_seg, loss_align, grad_norm_sq)."""
        self.lr_dev.fill_(float(lr))
        with ops.use_stream(torch.cuda.current_stream()):
            return self._step(images_s, label_s, images_t, None, regs_t)

    def capture(self, *a, **k):
        raise NotImplementedError('whole-step graph capture is provided for the SSL step only')

    def record_plan(self, *a, **k):
        raise NotImplementedError('plan replay is provided for the SSL step only (the stage-2 step has host actions that '
                                  'are not marked for recording)')

    def _step(self, images_s, label_s, images_t, soft_t, regs_t):
        m = self.model
        if not m.training:
            m.train()
        m._maybe_sync()
        m.flat_g.zero_()
        nb = images_s.shape[0]
        T = m.new_tape(groups=2)
        main = torch.cuda.current_stream()
        x1, x2, feat = m._forward_plan([images_s.contiguous().float(), images_t.contiguous().float()], T)
        s1, t1, s2, t2 = x1[:nb], x1[nb:], x2[:nb], x2[nb:]
        feat_s, feat_t = feat[:nb], feat[nb:]
        # ema-updating prototypes comes first here (train_align_reg.py:157): the target branch sees the new ones
        if self.reducer.active:
            # data-parallel ranks (or RGDA_FORCE_DDP at world 1, as in SSLStep): all-reduce the sufficient statistics
            # (per-class feature sums, pixel counts) and apply the totals -- the prototypes of the concatenated global
            # batch, identical on every rank.  Through the step's communicator when it has one (`comm=`: the torch-free
            # route), else through torch.distributed on `process_group`
            self.proto_stats, label_s_down = ops.proto_stats(feat_s, label_s, 16, self.ig, 0.75, self.C, stats=self.proto_stats)
            all_reduce_prototype_statistics(self.proto_stats, self.C, self.prototypes.shape[1], self.group, self.comm,
                                            world=self.world)
            ops.proto_apply(self.prototypes, self.proto_stats, self.pdecay)
        else:
            label_s_down = ops.proto_update(feat_s, label_s, self.prototypes, 16, self.ig, 0.75, self.pdecay)
        soft_t = ops.teacher_probs(t1, t2, tuple(images_t.shape[-2:]))             # :164-166
        if self.refine_label:
            soft, cm = ops.label_refine(feat_t, self.prototypes, t1, t2, soft_t, self.temp, return_ws=True)
            hard = ops.pseudo_select(soft, self.top, self.low, self.ig, classmax_ws=cm, check=False)
        else:
            hard = ops.pseudo_select(soft_t, self.top, self.low, self.ig, check=False)
        if self.sam_refine:
            regs = regs_t.squeeze(1) if regs_t.dim() == 4 else regs_t
            self._lrh_flag_off = (regs.shape[0] * self.max_regions * (self.C + 1)) * 4
            need = self._lrh_flag_off + 16
            if self.lrh_ws is None or self.lrh_ws.numel() < need:
                self.lrh_ws = torch.empty(need, dtype=torch.uint8, device=m.device)
            hard = ops.lrh(hard, regs.contiguous(), self.percent, self.C, self.ig, self.max_regions, check=False,
                           ws=self.lrh_ws)
        label_t = self._downscale(hard)                                              # aligner.downscale_gt, :180
        # ---- losses and their gradients
        loss_seg, gs1, gs2 = ops.upsample_ce(s1, s2, label_s, self.ig, None, True)
        n, k, h, w = feat.shape
        gfeat = torch.empty(n * h * w, k, dtype=BF, device=m.device)
        self.loss_align.zero_()
        ops.pcl_loss(feat_s, label_s_down, self.prototypes, self.pcl_temp, self.ig, 0.5, loss=self.loss_align,
                     dfeat=gfeat[:nb * h * w])
        ops.pcl_loss(feat_t, label_t, self.prototypes, self.pcl_temp, self.ig, 0.5, loss=self.loss_align,
                     dfeat=gfeat[nb * h * w:])
        zero = torch.zeros_like(gs1)
        self._backward_and_update(T, main, torch.cat([gs1, zero]), torch.cat([gs2, zero]), gfeat=gfeat)
        self.last_hard, self.last_label_t, self.last_label_s_down = hard, label_t, label_s_down
        return loss_seg, self.loss_align, self.gn

    def _downscale(self, hard):
        b, H, W = hard.shape
        dummy = torch.zeros((b, 4, H // 16, W // 16), device=hard.device)
        return ops.proto_update(dummy, hard, torch.zeros((self.C, 4), device=hard.device), 16, self.ig, 0.75, 0.5)
