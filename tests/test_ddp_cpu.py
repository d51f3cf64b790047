"""CPU: the N>1 gradient exchange (regda_amd/ddp.py) with the gloo backend, world_size 2, plus the
bucket planner.  The same class drives RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from regda_amd.ddp import FlatGradReducer, make_buckets


def test_bucket_plan_covers_buffer_in_reverse():
    b = make_buckets([100, 300, 350, 900], 1000, 200)
    assert b[0] == (900, 1000) or b[0][1] == 1000
    # contiguous, descending, complete
    assert b[-1][0] == 0
    for (a0, a1), (b0, b1) in zip(b, b[1:]):
        assert b1 == a0 and a0 < a1
    assert sum(e - s for s, e in b) == 1000
    assert make_buckets([], 10, 4) == [(0, 10)]
    # every bucket except the last-issued reaches the minimum size
    assert all(e - s >= 200 for s, e in b[:-1])
    # the last-issued bucket (first layers: final only when backward ends) is kept short when a cut allows it
    b = make_buckets([100, 300, 350, 900], 1000, 200, tail_elems=120)
    assert b[-1] == (0, 100) and b[-2][0] == 100 and sum(e - s for s, e in b) == 1000
    for (a0, a1), (b0, b1) in zip(b, b[1:]):
        assert b1 == a0 and a0 < a1
    assert make_buckets([100, 300], 1000, 250, tail_elems=50) == [(300, 1000), (0, 300)]      # no cut that short
    # ResNet-101-like layout (elements): stem + layer1 + layer2 = 1.44 M, layer3 blocks of 1.1 M, layer4, two heads
    cuts = [9408, 225000, 1440000] + [1440000 + 1117000 * i for i in range(1, 24)] + [27000000, 42000000, 65000000]
    b = make_buckets(cuts, 88653900, 12 << 20)
    assert b[-1] == (0, 1440000) and all(e - s >= (12 << 20) for s, e in b[:-2])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n = 10000
    g = torch.Generator().manual_seed(rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = FlatGradReducer(flat, [1000, 2500, 6000, 9000], bucket_elems=2000)
    red.reset()
    # backward progresses from the end of the buffer to the start
    for off in (9000, 6000, 2500, 1000):
        red.ready_down_to(off)
    red.finish()
    others = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(others, mine)
    expect = sum(others)
    ok = torch.allclose(flat, expect, rtol=1e-6, atol=1e-6)
    # the scale folded into the optimizer turns the sum into the mean
    ok = ok and abs(red.gscale - 1.0 / world) < 1e-12
    # a second step reuses the reducer
    flat.copy_(mine)
    red.reset()
    red.finish()
    ok = ok and torch.allclose(flat, expect, rtol=1e-6, atol=1e-6)
    # ClassBalance: per-class counts are summed over the ranks before the frequency EMA (regda_amd/gast/balance.py)
    from regda_amd.gast.balance import sync_class_counts
    cnt = torch.tensor([1.0, 2.0, 3.0]) * (rank + 1)
    sync_class_counts(cnt)
    ok = ok and torch.equal(cnt, torch.tensor([1.0, 2.0, 3.0]) * sum(r + 1 for r in range(world)))
    q.put((rank, bool(ok), len(red.buckets)))
    dist.destroy_process_group()


def _worker_bf16(rank, world, port, q):
    """payload='bf16' against the fp32 all-reduce on the same gradients: every element within the rounding the scheme
    states (half a bf16 ulp per contribution + half an ulp of the result), identical bits on both ranks, and exact where
    the inputs are exactly representable."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n = 10007                                           # not a multiple of world * 8: the shard padding is exercised
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(n, generator=g) * torch.logspace(-6, 2, n)        # eight decades of magnitudes
    bounds = [1000, 2500, 6000, 9000]
    ref = mine.clone()
    r32 = FlatGradReducer(ref, bounds, bucket_elems=2000)
    r32.reset(); r32.finish()
    ok = True
    for rep in range(2):                                # the reducer (and its staging buffers) are reused every step
        got = mine.clone()
        r16 = FlatGradReducer(got, bounds, bucket_elems=2000, payload='bf16') if rep == 0 else r16
        r16.flat_g = got
        r16.reset()
        for off in (9000, 6000, 2500, 1000):
            r16.ready_down_to(off)
        r16.finish()
        parts = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(parts, mine)
        # one bf16 ulp of a value v is 2^-7 |v| at most (8 significant bits): half an ulp per contribution + half of the sum's
        bound = sum(p.abs() for p in parts) * 2.0 ** -8 + ref.abs() * 2.0 ** -8 + 1e-30
        ok = ok and bool(((got - ref).abs() <= bound).all())
        both = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(both, got)
        ok = ok and torch.equal(both[0], both[1])       # the same bits on every rank
    # exactly representable inputs (small integers): the bf16 path is exact
    ints = torch.arange(n).remainder(64).float() * (rank + 1)
    e16 = ints.clone()
    rr = FlatGradReducer(e16, bounds, bucket_elems=2000, payload='bf16')
    rr.reset(); rr.finish()
    ok = ok and torch.equal(e16, torch.arange(n).remainder(64).float() * sum(r + 1 for r in range(world)))
    q.put((rank, bool(ok), len(r16.buckets)))
    dist.destroy_process_group()


def test_bf16_payload_matches_the_fp32_all_reduce_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def test_flat_grad_reducer_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(nb >= 3 for _, _, nb in res)


def _guarded(target, rank, world, port, q):
    try:
        target(rank, world, port, q)
    except BaseException as e:          # a worker that dies must not leave the parent waiting for the queue
        import traceback
        q.put((rank, False, 'worker failed: ' + traceback.format_exc()[-1500:]))
        raise e


def _spawn(target, world=2, timeout=300):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_guarded, args=(target, r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=timeout) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    return sorted(res)


def _worker_prototypes(rank, world, port, q):
    """ranks exchange the SUFFICIENT STATISTICS of update_prototype; the result is the reference's update on
    the concatenated batch (regda/gast/alignment.py:300-327), also for a class only one rank sees."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import numpy as np
    from oracle import labelpath as opath, labels as olab
    from regda_amd.ddp import all_reduce_prototype_statistics
    C, K, b, h, w = 6, 96, 2, 4, 4
    g = torch.Generator().manual_seed(100)
    protos = torch.randn(C, K, generator=g)

    def batch(r):
        gr = torch.Generator().manual_seed(200 + r)
        feat = torch.randn(b, K, h, w, generator=gr)
        cells = torch.randint(-1, C, (b, h, w), generator=gr)
        # class 3 only on rank 1, class 4 on nobody, class 5 only on rank 0 (a single cell)
        cells[cells == 4] = 0
        cells[cells == (3 if r == 0 else 5)] = 1
        if r == 0:
            cells[0, 0, 0] = 5
        label = cells.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()      # (b, 16h, 16w): pure cells
        return feat, label
    feat, label = batch(rank)
    ds = torch.from_numpy(olab.downscale_label(label.numpy(), 16, C, -1, 0.75))
    sums, cnt = opath.prototype_statistics(feat, ds, C, -1)
    stats = torch.cat([sums.reshape(-1), cnt, torch.zeros(4)]).contiguous()             # the layout rgda_proto_stats leaves
    all_reduce_prototype_statistics(stats, C, K)
    new = opath.apply_prototype_statistics(protos, stats[:C * K].view(C, K), stats[C * K:C * K + C], 0.996)
    # the reference on the concatenated global batch
    fs, ls = zip(*[batch(r) for r in range(world)])
    want, _ = opath.update_prototype(torch.cat(fs), torch.cat(ls), protos, 0.996, C, -1)
    ok = torch.allclose(new, want, rtol=1e-6, atol=1e-6)
    ok = ok and torch.equal(new[4], 0.996 * protos[4] + (1.0 - 0.996) * protos[4])     # a class nobody has: the old prototype
    # what the round-4 build did (average of per-rank updates) is a different number for the one-rank classes
    mine, _ = opath.update_prototype(feat, label, protos, 0.996, C, -1)
    avg = mine.clone()
    dist.all_reduce(avg)
    avg /= world
    differs = not torch.allclose(avg[3], want[3], rtol=1e-4, atol=1e-6) and not torch.allclose(avg[5], want[5], rtol=1e-4, atol=1e-6)
    # identical bits on every rank
    others = [torch.zeros_like(new) for _ in range(world)]
    dist.all_gather(others, new)
    same = all(torch.equal(o, new) for o in others)
    q.put((rank, bool(ok), bool(differs), bool(same)))
    dist.destroy_process_group()


def test_prototype_statistics_exchange_equals_the_global_batch_gloo_world2():
    res = _spawn(_worker_prototypes)
    assert all(len(r) == 4 and r[1] and r[2] and r[3] for r in res), res


def _worker_overlap(rank, world, port, q):
    """The overlap bookkeeping with real data: two ranks run the oracle's backward on two DIFFERENT batches; the flat
    gradient is filled in the order backward finalises it and `ready_down_to` is called at exactly the offsets
    `Deeplabv2._backward_plan` reports (Encoder.backward_progress_offsets), buckets cut at the model's boundaries.  A bucket
    issued before its gradients are final would reduce the NaN poison."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(4)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import model as omodel
    from oracle.step import CpuStep
    from regda_amd.models import Encoder as E
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    E.LAYERS.setdefault(rt, omodel.LAYERS[rt])
    lay = E.flat_layout(rt, 'ppm', 6)
    total = lay.pop('__total__')[0]
    sd = omodel.init_state_dict(rt, 6, seed=3)
    b = make_batch(b=2, size=64, seed=40 + rank, device='cpu')                 # a different batch on every rank
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(0))
    masks = (torch.ones(2, 512), torch.ones(2, 512))
    out = CpuStep(sd, protos, rt).step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], masks, masks)

    def phys(name, g):                                           # conv weights lie [Cout][kh][kw][Cin] in the flat buffers
        return (g.permute(0, 2, 3, 1) if g.dim() == 4 else g).reshape(-1)
    single = torch.zeros(total)
    for name, (off, n) in lay.items():
        single[off:off + n] = phys(name, out['grads'][name])
    flat = torch.zeros(total)
    for name, (off, n) in lay.items():
        flat[off:off + n] = float('nan')                         # not produced yet
    red = FlatGradReducer(flat, E.bucket_boundaries(rt, 'ppm', 6), bucket_elems=total // 7)
    red.reset()
    pending = sorted(lay.items(), key=lambda kv: -kv[1][0])      # backward finalises the buffer from its END
    issued_at = []
    for off in E.backward_progress_offsets(rt, 'ppm', 6):
        while pending and pending[0][1][0] >= off:
            name, (o, n) = pending.pop(0)
            flat[o:o + n] = single[o:o + n]
        before = red._next
        red.ready_down_to(off)
        issued_at += [(off, red.buckets[i]) for i in range(before, red._next)]
    for name, (o, n) in pending:                                 # the stem: final when backward returns
        flat[o:o + n] = single[o:o + n]
    red.finish()
    singles = [torch.zeros(total) for _ in range(world)]
    dist.all_gather(singles, single)
    ok = torch.equal(flat, singles[0] + singles[1])              # fp32 all-reduce of two ranks: bit for bit
    ok = ok and torch.equal(flat * red.gscale, (singles[0] + singles[1]) * 0.5)          # the mean the optimizer applies
    ok = ok and not torch.equal(singles[0], singles[1]) and bool(torch.isfinite(flat).all())
    ok = ok and all(a >= off for off, (a, _) in issued_at) and len(issued_at) >= 2        # something WAS issued during backward
    q.put((rank, bool(ok), len(red.buckets), len(issued_at)))
    dist.destroy_process_group()


def test_bucketed_reduce_at_the_real_backward_offsets_gloo_world2():
    res = _spawn(_worker_overlap)
    assert all(len(r) == 4 and r[1] for r in res), res
    assert all(r[2] >= 4 for r in res), res


def test_single_process_reducer_is_a_noop():
    flat = torch.arange(10.0)
    red = FlatGradReducer(flat, [5], bucket_elems=2)
    red.reset()
    red.ready_down_to(5)
    red.finish()
    assert torch.equal(flat, torch.arange(10.0)) and red.gscale == 1.0
    from regda_amd.gast.balance import sync_class_counts
    cnt = torch.tensor([4.0, 5.0])
    assert sync_class_counts(cnt) is cnt and torch.equal(cnt, torch.tensor([4.0, 5.0]))


def test_bench_launcher_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` without a launcher environment re-executes itself under torch.distributed.run with
    two ranks (here: the gloo self-test leg, no GPU); a launcher that started a different number of ranks is an error;
    host_cores() respects the affinity mask."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['sum'] == 3.0
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo'],
                       env=dict(env, WORLD_SIZE='1', RANK='0'), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr
    sys.path.insert(0, root)
    import bench
    n, info = bench.host_cores()
    assert 1 <= n <= info['affinity_cpus'] and n <= info['physical_cores_per_socket']
