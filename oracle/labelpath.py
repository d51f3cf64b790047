"""Floating-point label path of the SSL step, restated with stock PyTorch CPU fp32.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  pearson_dist / label_refine <- regda/gast/alignment.py:194-265, 283-298, 396-423
  local_prototypes / ema      <- regda/gast/alignment.py:300-327, 435-438
  cross_entropy_mean / loss_calc <- regda/gast/balance.py:88-101; regda/utils/tools.py:240-254
  class_balance_*             <- regda/gast/balance.py:27-60
  lr_at                       <- regda/utils/tools.py:191-207; tools/train_ssl_reg.py:85-87
  ema_update                  <- regda/utils/ema.py:46-51
  teacher_probs               <- regda/models/Encoder.py:152-155
"""
import torch
import torch.nn.functional as F

from . import labels as _lab

EPS = 1e-7  # alignment.py:44


def pearson_dist(feat1, feat2):
    """(n,k),(m,k) -> (n,m) in [0,1].  alignment.py:396-423.
    The reference materialises an (n,m,k) broadcast product; the sum over k of
    centred products is the same contraction written as a matmul here (fp32
    summation order differs, see tolerance in tests)."""
    k = feat1.shape[-1]
    c1 = feat1 - feat1.mean(dim=-1, keepdim=True)
    c2 = feat2 - feat2.mean(dim=-1, keepdim=True)
    cov = (c1.unsqueeze(1) * c2.unsqueeze(0)).sum(dim=-1) if feat1.shape[0] * feat2.shape[0] * k <= (1 << 26) \
        else c1 @ c2.t()
    cov = cov / (k - 1 + EPS)
    s1 = feat1.std(dim=-1).unsqueeze(1)
    s2 = feat2.std(dim=-1).unsqueeze(0)
    return (-1.0 * cov / (s1 * s2 + EPS) + 1.0) * 0.5


def softmax_T(x, temp=1.0, dim=1):
    return torch.softmax(x / temp, dim=dim)       # alignment.py:283-286


def superpixel_weight(label_t_sup, label_t_soft, temp):
    """alignment.py:238-253 -> (sup_weight (b,c,H,W), ignored (b,1,H,W) bool).  The reference's
    torch_scatter.scatter(src, index, dim=1, reduce='max') (third party, requirement.txt: torch-scatter, not vendored) is the
    per-(image, superpixel, class) maximum over the pixels carrying that id; entries no pixel indexes are never gathered."""
    b, c, H, W = label_t_soft.shape
    ids = label_t_sup.reshape(b, -1).long()                                # (b, HW)
    sup_cnt = int(ids.max())                                               # :241, over the whole batch
    src = label_t_soft.permute(0, 2, 3, 1).reshape(b, -1, c)               # (b, HW, c)
    table = torch.full((b, sup_cnt + 1, c), float('-inf'))
    table.scatter_reduce_(1, ids.unsqueeze(-1).expand(-1, -1, c), src, reduce='amax')          # :245
    prob = torch.gather(table, 1, ids.unsqueeze(-1).expand(-1, -1, c))     # :248
    prob = prob.reshape(b, H, W, c).permute(0, 3, 1, 2)
    prob = softmax_T(prob, temp, 1)                                        # :252
    return prob / (prob.max(dim=1, keepdim=True)[0] + 1e-7), (ids == sup_cnt).reshape(b, 1, H, W)      # :253, :242


def label_refine(feat_t, prototypes, preds_t, label_t_soft, refine=True, mode='all', temp=2.0, label_t_sup=None):
    """alignment.py:194-265; label_t_sup=None is what train_ssl_reg.py:214 passes."""
    if not refine:
        return label_t_soft
    b, k, h, w = feat_t.shape
    H, W = label_t_soft.shape[-2:]
    weight = 0
    if mode in ('all', 'p'):
        flat = feat_t.permute(0, 2, 3, 1).reshape(-1, k)
        simi = 1.0 / pearson_dist(flat, prototypes)                       # :216
        simi = simi.view(b, h, w, -1).permute(0, 3, 1, 2)
        simi = F.interpolate(simi, (H, W), mode='bilinear', align_corners=True)
        pw = softmax_T(simi, 1, 1)
        pw = pw / (pw.max(dim=1, keepdim=True)[0] + 1e-7)                # :222
        weight = weight + pw
    if mode in ('all', 'l'):
        if isinstance(preds_t, (list, tuple)):
            x1 = F.interpolate(preds_t[0], (H, W), mode='bilinear', align_corners=True)
            x2 = F.interpolate(preds_t[1], (H, W), mode='bilinear', align_corners=True)
            lw = (softmax_T(x1, temp, 1) + softmax_T(x2, temp, 1)) * 0.5     # :230-231
        else:
            lw = softmax_T(F.interpolate(preds_t, (H, W), mode='bilinear', align_corners=True), temp, 1)   # :233-234
        lw = lw / (lw.max(dim=1, keepdim=True)[0] + 1e-7)                # :235
        weight = weight + lw
    if label_t_sup is not None and mode in ('all', 's'):                 # superpixel view, :238-258
        sw, ignored = superpixel_weight(label_t_sup, label_t_soft, temp)
        if mode == 'all':
            weight = torch.where(ignored, weight, weight * sw)
        else:
            weight = torch.where(ignored, torch.ones_like(sw), sw)
    if isinstance(weight, int):
        return label_t_soft                                              # modes 's' / 'n' without superpixels, :260-261
    soft = weight * label_t_soft                                         # :263
    return soft / (soft.sum(dim=1, keepdim=True) + EPS)                  # :288-298


def local_prototypes(feat, label_ds, prototypes, class_num=6, ignore_label=-1):
    """alignment.py:300-321.  feat (b,k,h,w) f32, label_ds (b,1,h,w) int64."""
    b, k, h, w = feat.shape
    feats = feat.permute(0, 2, 3, 1).reshape(-1, k)
    lab = label_ds.reshape(-1).clone()
    lab[lab == ignore_label] = class_num
    onehot = F.one_hot(lab, class_num + 1)[:, :-1].to(feats.dtype)       # (n,c)
    n_inst = onehot.sum(0).unsqueeze(1).expand(class_num, k)
    local = (onehot.t() @ feats) / (n_inst + EPS)
    return torch.where(n_inst < 1, prototypes, local)


def update_prototype(feat_s, label_s, prototypes, decay=0.996, class_num=6, ignore_label=-1):
    """alignment.py:86-90 -> (new prototypes, downscaled label)."""
    ds = torch.from_numpy(_lab.downscale_label(label_s.numpy(), 16, class_num, ignore_label, 0.75))
    local = local_prototypes(feat_s, ds, prototypes, class_num, ignore_label)
    return (1.0 - decay) * local + decay * prototypes, ds               # :435-438


def prototype_statistics(feat, label_ds, class_num=6, ignore_label=-1):
    """The sufficient statistics of `local_prototypes` (alignment.py:300-317): (sums (c,k), counts (c,)).  They add over
    batches -- what data-parallel ranks exchange."""
    b, k, h, w = feat.shape
    feats = feat.permute(0, 2, 3, 1).reshape(-1, k)
    lab = label_ds.reshape(-1).clone()
    lab[lab == ignore_label] = class_num
    onehot = F.one_hot(lab, class_num + 1)[:, :-1].to(feats.dtype)
    return onehot.t() @ feats, onehot.sum(0)


def apply_prototype_statistics(prototypes, sums, counts, decay=0.996):
    """alignment.py:318-321 + :435-438 from (global) statistics: local = sums / (n + eps), old prototype where n < 1, EMA."""
    n_inst = counts.unsqueeze(1).expand_as(sums)
    local = torch.where(n_inst < 1, prototypes, sums / (n_inst + EPS))
    return (1.0 - decay) * local + decay * prototypes


def class_balance_local_freq(label, class_num=6, ignore_label=-1):
    """balance.py:45-53."""
    lab = label.reshape(-1)
    valid = lab != ignore_label
    local_cnt = valid.float().sum()
    cnt = torch.bincount(lab[valid], minlength=class_num).float()[:class_num]
    return cnt / (local_cnt + 1e-7)


def class_balance_weights(freq, temperature):
    """balance.py:37-43."""
    p = torch.softmax((1.0 - freq) / temperature, dim=0)
    return p / (p.max() + 1e-7)


class ClassBalanceState:
    """balance.py:15-60.  Stateful: `freq` is EMA-updated on EVERY call of
    get_class_weight_4pixel, i.e. once per head inside loss_calc(multi=True)."""

    def __init__(self, class_num=6, ignore_label=-1, decay=0.99, temperature=0.5):
        self.C, self.ig, self.decay, self.T = class_num, ignore_label, decay, temperature
        self.freq = torch.ones(class_num) / class_num                     # :25

    def pixel_weight(self, label):
        local = class_balance_local_freq(label, self.C, self.ig)
        self.freq = (1.0 - self.decay) * local + self.decay * self.freq   # :34-35
        w = class_balance_weights(self.freq, self.T)
        lab = label.reshape(-1)
        return torch.where(lab != self.ig, w[lab.clamp(min=0)], torch.zeros(()))  # :29-32


def cross_entropy_mean(pred, label, ignore_label=-1, balancer=None):
    """balance.py:88-101: CE(reduction='none', ignore_index) -> mean over ALL pixels."""
    loss = F.cross_entropy(pred, label, ignore_index=ignore_label, reduction='none').view(-1)
    if balancer is not None:
        loss = loss * balancer.pixel_weight(label)
    return loss.mean()


def loss_calc(preds, label, ignore_label=-1, balancer=None):
    """tools.py:240-254 (multi=True): bilinear(ac=True) up, CE, mean over heads."""
    loss = 0
    for p in preds:
        if p.shape[-2:] != label.shape[-2:]:
            p = F.interpolate(p, size=label.shape[-2:], mode='bilinear', align_corners=True)
        loss = loss + cross_entropy_mean(p, label.long(), ignore_label, balancer)
    return loss / len(preds)


def lr_at(i_iter, base_lr=1e-2, stage3_steps=6000, power=0.9):
    """tools.py:191-207 with NUM_STEPS = 1.5*STAGE3, PREHEAT = STAGE3/20
    (train_ssl_reg.py:85-87)."""
    num_steps = stage3_steps * 1.5
    preheat = int(stage3_steps / 20)
    if i_iter < preheat:
        return base_lr * (float(i_iter) / preheat)
    return base_lr * ((1 - float(i_iter) / num_steps) ** power)


def ema_update(shadow, param, decay):
    """ema.py:46-51."""
    return (1.0 - decay) * param + decay * shadow


def teacher_probs(x1, x2, size):
    """Encoder.py:152-155."""
    x1 = F.interpolate(x1, size, mode='bilinear', align_corners=True)
    x2 = F.interpolate(x2, size, mode='bilinear', align_corners=True)
    return (x1.softmax(dim=1) + x2.softmax(dim=1)) / 2


def prototype_contrastive_loss(protos, feat, labels, temperature=8.0, ignore_label=-1):
    """regda/loss.py:10-47 (PrototypeContrastiveLoss.forward): drop ignored pixels, L2-normalise features and
    prototypes (tnf.normalize, eps 1e-12), logits = f . P^T / temperature, mean cross entropy."""
    import torch.nn.functional as F
    if feat.dim() != 2:
        k = feat.size(1)
        feat = feat.permute(0, 2, 3, 1).reshape(-1, k)
    labels = labels.reshape(-1)
    mask = labels != ignore_label
    labels, feat = labels[mask], feat[mask]
    feat = F.normalize(feat, p=2, dim=1)
    protos = F.normalize(protos, p=2, dim=1)
    logits = feat.mm(protos.permute(1, 0).contiguous()) / temperature
    return F.cross_entropy(logits, labels)
