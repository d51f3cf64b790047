"""dev: the BatchNorm element-wise passes at the geometries of the step, replayed below the ABI (no host time between
launches), rotating over enough buffers that the Infinity Cache does not serve the re-reads: us and TB/s of the
bytes each launch must move."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regda_amd import ops, plan
BF = torch.bfloat16
dev = 'cuda'
NREP = 8
REPS = 24


def timed(body):
    p = plan.Plan()
    p.record(body)
    p.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); p.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


GEOS = [(16384, 256, 0), (16384, 1024, 1), (16384, 512, 0), (16384, 2048, 1), (65536, 128, 0), (65536, 512, 1),
        (262144, 64, 0), (262144, 256, 1), (1048576, 64, 0)]
for (M, C, with_res) in GEOS:
    nb = max(2, min(REPS, int(700e6 // (M * C * 2 * 3))))
    xs = [torch.randn(M, C, device=dev).to(BF) for _ in range(nb)]
    rs = [torch.randn(M, C, device=dev).to(BF) for _ in range(nb)] if with_res else None
    ys = [torch.empty(M, C, dtype=BF, device=dev) for _ in range(nb)]
    mk = [torch.empty(M, C // 8, dtype=torch.uint8, device=dev) for _ in range(nb)]
    stats = torch.rand(2 * NREP * 2 * C, device=dev) * 100 + 2000.0
    stats.view(2, NREP, 2, C)[:, :, 0] = 0.1
    mi = torch.zeros(2, 2, C, device=dev); mi[:, 1] = 1
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    dga, dbe = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    sums = torch.rand(2 * NREP * 2 * C, device=dev)

    def fwd():
        for i in range(REPS):
            k = i % nb
            ops.bn_train_apply(xs[k], stats, mi, rm, rv, nbt, ga, be, ys[k], M, C, True, rs[k] if with_res else None,
                               groups=2, relu_mask=mk[k])

    def bwd():
        for i in range(REPS):
            k = i % nb
            ops.bn_bwd_apply(xs[k], None, ys[k], mi, ga, sums, ys[(k + 1) % nb], M, C, True, None, dga, dbe, groups=2,
                             relu_mask=mk[k])
    tf, tb = timed(fwd), timed(bwd)
    bf = M * C * 2 * (2 + with_res) + M * C // 8
    bb = M * C * 2 * 3 + M * C // 8
    print('M=%7d C=%4d res=%d: bn_train_apply %6.1f us %5.2f TB/s | bn_bwd_apply %6.1f us %5.2f TB/s' %
          (M, C, with_res, tf, bf / tf / 1e6, tb, bb / tb / 1e6), flush=True)
    del xs, rs, ys, mk
