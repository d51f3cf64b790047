"""One rank of the multi-rank RCCL exercise of tests/test_ddp_gpu.py::test_multi_rank_rccl_step (not collected by pytest).

Launched as `python -m torch.distributed.run --nproc-per-node N tests/ddp_rank_worker.py`, one rank per GPU.  For every
(gradient payload, who issues the collectives) it runs ONE SSL step of the shallow test topology on a per-rank batch and checks,
on every rank:
  * the exchanged flat gradient == the sum over ranks of the gradients the same step leaves with the exchange switched off
    (fp32 payload: bit for bit at N <= 2, where the sum has one association; otherwise and for bf16 within the payload's rounding);
  * weights, EMA-free prototypes and the gradient are bit-identical on all ranks after the step (min == max of int64 checksums).
Rank 0 prints `MULTI_RANK_OK {...}` as its last line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_step(seed_batch, comm=None, grad_payload='fp32', exchange=True):
    from oracle import model as omodel
    from regda_amd.ddp import FlatGradReducer
    from regda_amd.models import Encoder as enc
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.ssl import SSLStep
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    enc.LAYERS.setdefault(rt, omodel.LAYERS[rt])
    m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                       cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                       inchannels=2048, num_classes=6, is_ins_norm=True))
    m.load_state_dict(omodel.init_state_dict(rt, 6, seed=2), strict=True)        # the same weights on every rank
    ones = torch.ones(4, 512)
    m.set_drop_masks(ones, ones)
    b = make_batch(b=2, size=128, seed=seed_batch)                               # a different batch per rank
    init = FlatGradReducer.__init__
    if not exchange:
        def local_only(self, *a, **k):
            init(self, *a, **k)
            self.world, self.force = 1, False
        FlatGradReducer.__init__ = local_only
    try:
        st = SSLStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(3)), bucket_elems=1 << 20,
                     comm=comm, grad_payload=grad_payload)
    finally:
        FlatGradReducer.__init__ = init
    out = st.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], lr=1e-3)
    torch.cuda.synchronize()
    return m, st, [float(x.item()) for x in out]


def bits(t):
    return t.contiguous().view(torch.int32).to(torch.int64).sum()


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
    dist.init_process_group('nccl')          # lazy form (no device_id=): see bench.py
    from regda_amd.ddp import RcclComm
    report = {'world': world}
    # the gradients each rank computes on its own, summed over the ranks by a plain collective: the reference
    m0, st0, _ = build_step(13 + rank, exchange=False)
    assert not st0.reducer.active
    ref = m0.flat_g.clone()
    dist.all_reduce(ref)
    protos_local = st0.prototypes.clone()
    for payload in ('fp32', 'bf16'):
        for route in ('torch', 'abi'):
            comm = RcclComm.from_torch_store() if route == 'abi' else None
            try:
                m1, st1, out = build_step(13 + rank, comm=comm, grad_payload=payload)
                assert st1.world == world and (st1.reducer.active or world == 1)
                if world > 1:
                    assert st1.reducer._next == len(st1.reducer.buckets) >= 3       # every bucket was exchanged
                g = m1.flat_g
                if world == 1:
                    assert torch.equal(g, ref), (payload, route)
                elif payload == 'fp32' and world == 2:
                    assert torch.equal(g, ref), (payload, route, float((g - ref).abs().max()))
                else:
                    rel = float((g - ref).norm() / ref.norm())
                    assert rel < (1e-6 if payload == 'fp32' else 6e-3), (payload, route, rel)
                # every rank holds the same bits afterwards
                parts = torch.stack([bits(m1.flat_p), bits(m1.flat_g), bits(st1.prototypes)])
                lo, hi = parts.clone(), parts.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                assert torch.equal(lo, hi), (payload, route, lo.tolist(), hi.tolist())
                if world > 1:       # the prototypes are those of the GLOBAL batch, not this rank's
                    assert not torch.equal(st1.prototypes, protos_local)
                report[f'{payload}/{route}'] = {'loss': out[:2], 'checksum': int(lo[0])}
            finally:
                torch.cuda.synchronize()
                if comm is not None:
                    comm.destroy()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print('MULTI_RANK_OK ' + json.dumps(report), flush=True)


if __name__ == '__main__':
    main()
