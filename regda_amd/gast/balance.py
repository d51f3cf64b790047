"""CrossEntropy + ClassBalance -- mirror of regda/gast/balance.py:15-101 (the "CE + reweight" of the
SSL path).  The loss is evaluated by the fused bilinear-upsample + CE kernel (`rgda_upsample_ce`)."""
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import ops


def sync_class_counts(cnt, group=None, comm=None):
    """Data-parallel runs: the per-class pixel counts of one step are summed over the ranks before the
    frequency EMA, so every rank carries the SAME `freq` (and class weights) -- the statistic of the global batch.
    With one process this is the reference's single-GPU arithmetic unchanged."""
    if comm is not None:                    # a regda_amd.ddp.RcclComm: the library's own RCCL entry point (fp32 counts)
        if comm.world > 1:
            comm.all_reduce(cnt)
    elif dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return cnt


class ClassBalance(nn.Module):
    def __init__(self, class_num=7, ignore_label=-1, decay=0.99, temperature=0.5, process_group=None, comm=None):
        super().__init__()
        assert temperature > 0
        self.class_num, self.ignore_label = class_num, ignore_label
        self.process_group = process_group
        self.comm = comm            # regda_amd.ddp.RcclComm instead of torch.distributed
        self.decay, self.temperature, self.eps = decay, temperature, 1e-7
        self.freq = torch.ones([class_num], device='cuda').float() / class_num

    def ema_update(self, label):
        cnt = ops.class_count(label, self.class_num).float()          # per-class pixel counts
        cnt = sync_class_counts(cnt.contiguous(), self.process_group, self.comm)
        local = cnt / (cnt.sum() + self.eps)                           # balance.py:45-53
        self.freq = (1.0 - self.decay) * local + self.decay * self.freq

    def _get_class_wight(self):
        p = torch.softmax((1.0 - self.freq) / self.temperature, dim=0)
        return p / (p.max() + self.eps)

    def next_class_weight(self, label):
        """freq EMA update + class weights: what get_class_weight_4pixel (balance.py:27-35) does,
        returned per class (the kernel looks the pixel's class up)."""
        self.ema_update(label)
        return self._get_class_wight().detach()

    def __str__(self):
        f = self.freq.cpu().numpy()
        w = self._get_class_wight().cpu().numpy()
        return ('class frequency: ' + ', '.join(f'{v:.3f}' for v in f) +
                ';\tselect probability: ' + ', '.join(f'{v:.3f}' for v in w))


class _UpCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, label, ignore_label, class_weight):
        loss, g1, g2 = ops.upsample_ce(p1, p2, label, ignore_label, class_weight, want_grad=True)
        ctx.save_for_backward(g1, g2)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        g1, g2 = ctx.saved_tensors
        return g1 * go, g2 * go, None, None, None


class CrossEntropy(nn.Module):
    def __init__(self, ignore_label=-1, class_balancer=None):
        super().__init__()
        self.ignore_label = ignore_label
        self.class_balancer = class_balancer

    def _weights(self, labels, heads):
        if self.class_balancer is None:
            return None
        # the balancer is EMA-updated once per head (balance.py:27-28 is called inside each loss_fn call)
        ws = [self.class_balancer.next_class_weight(labels) for _ in range(heads)]
        if heads == 1:
            ws = ws * 2
        return torch.stack(ws, 0)

    def forward(self, preds, labels):
        """preds [B,C,H,W] logits (any resolution), labels [B,H,W] -> mean over ALL pixels."""
        cw = self._weights(labels, 1)
        return _UpCE.apply(preds, preds, labels.long(), self.ignore_label, cw)

    def forward_multi(self, preds, labels):
        """loss_calc(multi=True) in one fused pass: mean over the two heads."""
        assert len(preds) == 2
        cw = self._weights(labels, 2)
        return _UpCE.apply(preds[0], preds[1], labels.long(), self.ignore_label, cw)
