"""GPU: the RCCL path of the data-parallel step on ONE GPU.  `torch.distributed` backend "nccl" (= RCCL) is initialised
at world size 1 and RGDA_FORCE_DDP=1 makes the step issue its bucketed, stream-overlapped all-reduces exactly as it
does at world size N (regda_amd/ddp.py, regda_amd/ssl.py): with one rank every all-reduce is the identity, so the
step must reproduce the plain step.  (The N > 1 exchange itself is covered by tests/test_ddp_cpu.py with gloo; an 8-GPU
run is the driver's.)"""
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import model as omodel

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def nccl_world1(monkeypatch):
    monkeypatch.setenv('RGDA_FORCE_DDP', '1')
    # (no device_id=: see bench.py -- the eager, device-bound form slows every kernel of the process down)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1)
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _run(steps, bucket_elems, overlap_comm=True, class_balance=False, comm=None, grad_payload='fp32'):
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.ssl import SSLStep
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                       cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                       inchannels=2048, num_classes=6, is_ins_norm=True))
    m.load_state_dict(omodel.init_state_dict(rt, 6, seed=2), strict=True)
    ones = torch.ones(4, 512)
    m.set_drop_masks(ones, ones)
    b = make_batch(b=2, size=128, seed=13)
    st = SSLStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(3)), bucket_elems=bucket_elems,
                 overlap_comm=overlap_comm, comm=comm, grad_payload=grad_payload)
    out = None
    for i in range(steps):
        out = st.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], lr=1e-3)
    torch.cuda.synchronize()
    return m, st, [float(x.item()) for x in out]


def _run_align(comm=None):
    from regda_amd.align import AlignStep
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                       cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                       inchannels=2048, num_classes=6, is_ins_norm=True))
    m.load_state_dict(omodel.init_state_dict(rt, 6, seed=2), strict=True)
    ones = torch.ones(4, 512)
    m.set_drop_masks(ones, ones)
    b = make_batch(b=2, size=128, seed=13)
    st = AlignStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(3)), bucket_elems=1 << 20, comm=comm)
    out = st.step(b['images_s'], b['label_s'], b['images_t'], b['regs_t'], lr=1e-3)
    torch.cuda.synchronize()
    return m, st, [float(x.item()) for x in out]


def test_forced_rccl_step_equals_plain_step(nccl_world1):
    from regda_amd.ddp import FlatGradReducer
    m1, st1, out1 = _run(1, bucket_elems=1 << 20)                   # small buckets: several all-reduces per step
    assert st1.reducer.force and st1.world == 1 and len(st1.reducer.buckets) >= 3
    assert st1.reducer._next == len(st1.reducer.buckets)            # every bucket was issued
    m2, st2, out2 = _run(1, bucket_elems=1 << 20, overlap_comm=False)   # one exchange after backward
    # reference: the same step with the reducer switched off
    force = FlatGradReducer.__init__

    def no_force(self, *a, **k):
        force(self, *a, **k)
        self.force = False
    FlatGradReducer.__init__ = no_force
    try:
        m0, st0, out0 = _run(1, bucket_elems=1 << 20)
    finally:
        FlatGradReducer.__init__ = force
    assert not st0.reducer.force
    # not bit-identical even between two plain runs: the fp32 BatchNorm statistics and weight gradients are summed
    # with atomics in a varying order and the net amplifies that last-bit noise (same bounds as
    # tests/test_model_gpu.py::test_grouped_forward_backward_equals_two_separate_passes)
    for got in (out1, out2):
        assert got[0] == pytest.approx(out0[0], rel=2e-2) and got[1] == pytest.approx(out0[1], rel=3e-2, abs=1e-3)
        assert got[2] == pytest.approx(out0[2], rel=6e-2)
    for mm in (m1, m2):
        cos = (mm.flat_g @ m0.flat_g / (mm.flat_g.norm() * m0.flat_g.norm())).item()
        assert cos > 0.97 and abs(mm.flat_g.norm().item() / m0.flat_g.norm().item() - 1) < 0.05     # (0.989 .. 0.998 seen)
        assert ((mm.flat_p - m0.flat_p).norm() / m0.flat_p.norm()).item() < 1e-4
    # the prototype exchange (statistics -> all-reduce -> apply) at one rank IS update_prototype: same kernels, same bits
    assert st1.proto_stats is not None and st0.proto_stats is None
    assert torch.equal(st1.prototypes, st0.prototypes)


def test_prototype_statistics_of_two_half_batches_give_the_global_batch_update():
    """SURVEY.md 8e through the C ABI: rgda_proto_stats on two halves of a batch, summed (what the all-reduce does), then
    rgda_proto_apply == rgda_proto_update on the whole batch == the oracle's update_prototype on it
    (regda/gast/alignment.py:300-327) -- including a class that only one half contains and a class nobody has."""
    from oracle import labelpath as opath
    from regda_amd import ops
    g = torch.Generator().manual_seed(5)
    b, K, h, w, C = 4, 2048, 8, 8, 6
    feat = torch.randn(b, K, h, w, generator=g).cuda()
    cells = torch.randint(-1, C, (b, h, w), generator=g)
    cells[cells == 4] = 0                       # class 4: nobody
    cells[:2][cells[:2] == 3] = 1               # class 3: second half only
    label = cells.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous().cuda()
    protos0 = torch.randn(C, K, generator=g).cuda()
    whole = protos0.clone()
    ds = ops.proto_update(feat, label, whole, 16, -1, 0.75, 0.996)
    # stats + apply on the whole batch: the same two kernels, the same bits
    st, ds2 = ops.proto_stats(feat, label)
    one = protos0.clone()
    ops.proto_apply(one, st, 0.996)
    assert torch.equal(one, whole) and torch.equal(ds, ds2)
    # two ranks' halves
    s0, _ = ops.proto_stats(feat[:2].contiguous(), label[:2].contiguous())
    s1, _ = ops.proto_stats(feat[2:].contiguous(), label[2:].contiguous())
    n = C * K + C
    tot = s0.clone()
    tot[:n] = s0[:n] + s1[:n]
    assert float(s0[C * K + 3]) == 0 and float(s1[C * K + 3]) > 0 and float(tot[C * K + 4]) == 0
    two = protos0.clone()
    ops.proto_apply(two, tot, 0.996)
    torch.testing.assert_close(two, whole, rtol=1e-6, atol=1e-6)
    want, _ = opath.update_prototype(feat.cpu(), label.cpu(), protos0.cpu(), 0.996, C, -1)
    torch.testing.assert_close(two.cpu(), want, rtol=1e-5, atol=1e-6)
    assert torch.equal(two[4], whole[4])
    # the average of per-rank updates (the round-4 exchange) is NOT that number for the one-rank class
    r0, r1 = protos0.clone(), protos0.clone()
    ops.proto_update(feat[:2].contiguous(), label[:2].contiguous(), r0, 16, -1, 0.75, 0.996)
    ops.proto_update(feat[2:].contiguous(), label[2:].contiguous(), r1, 16, -1, 0.75, 0.996)
    assert not torch.allclose(0.5 * (r0 + r1)[3], whole[3], rtol=1e-4, atol=1e-6)


def test_class_balance_counts_go_through_the_process_group(nccl_world1):
    from regda_amd.gast.balance import ClassBalance
    lab = torch.randint(-1, 6, (2, 64, 64), device='cuda')
    cb = ClassBalance(class_num=6, ignore_label=-1)
    w = cb.next_class_weight(lab)
    cnt = torch.stack([(lab == c).sum() for c in range(6)]).float()
    freq = 0.01 * cnt / (cnt.sum() + 1e-7) + 0.99 * torch.ones(6, device='cuda') / 6
    torch.testing.assert_close(cb.freq, freq, rtol=1e-6, atol=1e-7)
    assert w.shape == (6,) and float(w.max()) <= 1.0


def test_recorded_plan_replays_the_collectives(nccl_world1):
    """The multi-GPU step as a recorded launch plan (bench.py's default at every world size): the bucketed gradient
    all-reduces, their stream waits and the prototype-statistics all-reduce are host actions of the plan and are re-issued by every
    replay -- counted here through the reducer, and the replayed steps track the eager ones."""
    import torch.distributed as dist_mod
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.ssl import SSLStep
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    sd = omodel.init_state_dict(rt, 6, seed=2)
    ones = torch.ones(4, 512)
    b = make_batch(b=2, size=128, seed=13)
    calls = []
    real = dist_mod.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)

    def run(use_plan):
        m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                           cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                           inchannels=2048, num_classes=6, is_ins_norm=True))
        m.load_state_dict(sd, strict=True)
        m.set_drop_masks(ones, ones)
        st = SSLStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(3)), bucket_elems=1 << 20)
        outs = []
        for i in range(4):
            if use_plan and i == 1:
                st.record_plan(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'])
                outs.append([float(x.item()) for x in st._out])
                continue
            outs.append([float(x.item()) for x in st.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'],
                                                          b['regs_t'], lr=1e-3)])
        torch.cuda.synchronize()
        return m, st, outs

    dist_mod.all_reduce = counting
    try:
        m_e, st_e, out_e = run(False)
        n_eager = len(calls)
        del calls[:]
        m_p, st_p, out_p = run(True)
        n_plan = len(calls)
    finally:
        dist_mod.all_reduce = real
    per_step = len(st_e.reducer.buckets) + 1                    # gradient buckets + the prototype statistics
    assert per_step >= 4 and n_eager == 4 * per_step
    assert st_p._plan is not None and n_plan == n_eager         # every replay issued every collective again
    for oe, op in zip(out_e, out_p):
        assert op[0] == pytest.approx(oe[0], rel=5e-2) and op[2] == pytest.approx(oe[2], rel=0.3)
    assert ((m_p.flat_p - m_e.flat_p).norm() / m_e.flat_p.norm()).item() < 1e-3
    assert ((st_p.prototypes - st_e.prototypes).norm() / st_e.prototypes.norm()).item() < 5e-3


def test_bf16_payload_exchange_through_rccl(nccl_world1):
    """payload='bf16' on the GPU through RCCL at world size 1 (all-to-all, rgda_ddp_accumulate_bf16, all-gather,
    rgda_cast_f32 on the reducer's own stream): with one rank the exchanged gradient is the bf16-rounded gradient, bit for
    bit, and a full step with that payload stays within bf16 rounding of the plain step's losses."""
    from regda_amd.ddp import FlatGradReducer
    g = torch.randn(100003, device='cuda') * torch.logspace(-5, 3, 100003, device='cuda')
    want = g.to(torch.bfloat16).float()
    red = FlatGradReducer(g, [1000, 30000, 70000], bucket_elems=20000, payload='bf16')
    assert red.force and red.active
    red.reset()
    for off in (70000, 30000, 1000):
        red.ready_down_to(off)
    red.finish()
    torch.cuda.synchronize()
    assert torch.equal(g, want)
    g2 = want.clone()                    # a second step reuses the staging buffers; bf16 values pass unchanged
    red.flat_g = g2
    red.reset()
    red.finish()
    torch.cuda.synchronize()
    assert torch.equal(g2, want)
    # a whole step
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.ssl import SSLStep
    from regda_amd.synthetic import make_batch
    rt = 'resnet17t'
    outs = {}
    for payload in ('fp32', 'bf16'):
        m = Deeplabv2(dict(backbone=dict(resnet_type=rt, output_stride=16, pretrained=False), multi_layer=True,
                           cascade=False, use_ppm=True, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                           inchannels=2048, num_classes=6, is_ins_norm=True))
        m.load_state_dict(omodel.init_state_dict(rt, 6, seed=2), strict=True)
        ones = torch.ones(4, 512)
        m.set_drop_masks(ones, ones)
        b = make_batch(b=2, size=128, seed=13)
        st = SSLStep(m, torch.randn(6, 2048, generator=torch.Generator().manual_seed(3)), bucket_elems=1 << 20, grad_payload=payload)
        st.measure_comm = True
        for _ in range(2):
            out = st.step(b['images_s'], b['label_s'], b['images_t'], b['soft_t'], b['regs_t'], lr=1e-3)
        torch.cuda.synchronize()
        assert st.comm_events is not None and st.comm_events[0].elapsed_time(st.comm_events[1]) >= 0.0
        outs[payload] = ([float(x.item()) for x in out], m.flat_p.clone())
    (l32, p32), (l16, p16) = outs['fp32'], outs['bf16']
    assert l16[0] == pytest.approx(l32[0], rel=2e-3) and l16[1] == pytest.approx(l32[1], rel=2e-3, abs=2e-3)
    assert float((p16 - p32).norm() / p32.norm()) < 1e-4


def test_rccl_entry_points_of_the_c_abi(monkeypatch):
    """rgda_comm_* (include/rgda_hip.h; SURVEY.md 8b "RCCL wrappers for (e)") WITHOUT torch.distributed: a communicator of
    one rank -- every collective is the identity -- for each dtype the step exchanges, then the SSL step with its gradient
    buckets (fp32 and bf16 payload) and its prototype statistics routed through that communicator: it must reproduce the
    step whose reducer is switched off.  (N > 1: the same entry points with the id shipped by the host; the driver's 8-GPU run.)"""
    from regda_amd.ddp import FlatGradReducer, RcclComm
    assert not dist.is_initialized()
    comm = RcclComm(RcclComm.unique_id(), 0, 1)
    try:
        for dt in (torch.float32, torch.bfloat16, torch.int64, torch.float64):
            x = (torch.arange(4096, device='cuda') % 97).to(dt)
            ref = x.clone()
            comm.all_reduce(x)
            r1, r2 = torch.zeros_like(x), torch.zeros_like(x)
            comm.all_gather(x, r1)
            comm.all_to_all(x, r2)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            comm.all_reduce(x, stream=side)                      # an explicit stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(x, ref) and torch.equal(r1, ref) and torch.equal(r2, ref), dt
        with pytest.raises(ValueError):
            comm.all_to_all(x, x)                                # in place is refused (RGDA_ERR_ARG)
        monkeypatch.setenv('RGDA_FORCE_DDP', '1')
        runs = [_run(1, bucket_elems=1 << 20, comm=comm), _run(1, bucket_elems=1 << 20, comm=comm, grad_payload='bf16')]
        for m1, st1, out1 in runs:
            assert st1.reducer.comm is comm and st1.reducer.force and st1.reducer._next == len(st1.reducer.buckets) >= 3
        monkeypatch.delenv('RGDA_FORCE_DDP')
        m0, st0, out0 = _run(1, bucket_elems=1 << 20)
        assert not st0.reducer.active
        for (m1, st1, out1), wtol in zip(runs, (1e-4, 1e-4)):
            assert out1[0] == pytest.approx(out0[0], rel=2e-2) and out1[1] == pytest.approx(out0[1], rel=3e-2, abs=1e-3)
            cos = (m1.flat_g @ m0.flat_g / (m1.flat_g.norm() * m0.flat_g.norm())).item()
            assert cos > 0.97, cos
            assert ((m1.flat_p - m0.flat_p).norm() / m0.flat_p.norm()).item() < wtol
            assert torch.equal(st1.prototypes, st0.prototypes)
        # the stage-2 step takes the same route: its prototype statistics go through the communicator too (they used to
        # bypass it, and without torch.distributed silently stayed local)
        monkeypatch.setenv('RGDA_FORCE_DDP', '1')
        a1 = _run_align(comm=comm)
        assert a1[1].reducer.active and a1[1].comm is comm and a1[1].proto_stats is not None
        monkeypatch.delenv('RGDA_FORCE_DDP')
        a0 = _run_align()
        assert not a0[1].reducer.active and a0[1].proto_stats is None
        assert a1[2][0] == pytest.approx(a0[2][0], rel=2e-2) and a1[2][1] == pytest.approx(a0[2][1], rel=3e-2, abs=1e-3)
        assert torch.allclose(a1[1].prototypes, a0[1].prototypes, rtol=1e-5, atol=1e-6)
    finally:
        torch.cuda.synchronize()
        comm.destroy()


def test_prototype_statistics_refuse_to_stay_local_at_world_2():
    """No communicator, no process group, but the caller says there are two ranks: raise instead of applying local totals."""
    from regda_amd.ddp import all_reduce_prototype_statistics
    assert not dist.is_initialized()
    stats = torch.zeros(6 * 2048 + 6, device='cuda')
    all_reduce_prototype_statistics(stats, 6, 2048, world=1)
    with pytest.raises(RuntimeError):
        all_reduce_prototype_statistics(stats, 6, 2048, world=2)


@pytest.mark.parametrize('nproc', [1, 2])
def test_multi_rank_rccl_step(nproc):
    """A REAL N-rank RCCL step (one process per GPU, `torch.distributed.run`): tests/ddp_rank_worker.py runs the SSL step
    with both gradient payloads through torch.distributed and through the library's own rgda_comm_* entry points and
    asserts, on every rank, exchanged gradient == sum of the single-rank gradients and identical weight / gradient /
    prototype bits across ranks.  nproc = 1 runs everywhere (the worker's whole code path on one GPU); nproc = 2 needs
    two GPUs and is skipped on the single-GPU boxes of this pool -- the first multi-GPU box exercises it before any
    scaling run does."""
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < nproc:
        pytest.skip(f'{nproc} GPUs needed, {torch.cuda.device_count()} visible')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('RGDA_FORCE_DDP', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(root, 'tests', 'ddp_rank_worker.py')]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert any(l.startswith('MULTI_RANK_OK ') for l in r.stdout.splitlines()), r.stdout[-2000:]
