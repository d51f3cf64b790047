"""Headline benchmark: src+tgt 512x512 image-pairs/sec for one full RegDA self-training (SSL) step
(BASELINE.json metric; config st.regda.2potsdam, ResNet-101 DeepLabV2/PPM, batch 8+8 per GPU, bf16 MFMA
compute with fp32 accumulate, online EMA teacher) on N MI355X GPUs, one process per GPU over RCCL.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  A step = model(src), model(tgt), EMA-teacher forward on tgt, label_refine,
pseudo_selection, LRH, update_prototype, 2x loss, backward of both passes, grad all-reduce, clip + SGD +
EMA: nothing is skipped inside the timed region.  `roofline` is measured live with HIP events around
every conv launch of one extra (untimed) step; `cpu_baseline` times the CPU oracle (stock PyTorch fp32)
on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_PAIR_STUDENT = 1087.0     # (fwd + dgrad + wgrad) x (src + tgt) conv FLOPs
GFLOP_PER_PAIR_TEACHER = 181.17     # + one eval forward of the EMA teacher on the target image
MFMA_PEAK_TFLOPS = 2500.0       # bf16 dense, MI355X_MICROARCH.md
# what the matrix pipes sustain on N(0,1) bf16 operands with no data movement at all (2440 on all-zero operands):
# measured, scripts/dev/mfma_ceiling.py -> profiles/r02_mfma_ceiling.txt.  Reported next to the nominal peak only.
MFMA_SUSTAINED_TFLOPS = 1810.0
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None, help='number of ranks (one per GPU); without a launcher environment and N > 1 the script re-executes itself under torch.distributed.run')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="'gloo' = launcher self-test without GPUs (rendezvous + one all-reduce, no step)")
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--repeats', type=int, default=2, help='further timed windows of --steps steps after the reported one (spread of a short window on a fresh box): ms_per_step_windows in the line')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='source (= target) images per GPU')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--model', default='resnet101')
    ap.add_argument('--no-teacher', action='store_true', help='offline soft labels (the reference\'s mode) instead of the online EMA teacher')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--full-json', default=os.path.join(ROOT, 'gpurun_out', 'bench_full.json'),
                    help='where the FULL record goes (per-kernel roofline table, timing windows, host probes, CPU-baseline '
                         'phases); stdout carries one short line with the contract keys, roofline and cpu_baseline')
    ap.add_argument('--serial', action='store_true', help='one stream only (clean per-kernel profiles)')
    ap.add_argument('--grad-payload', default='fp32', choices=['fp32', 'bf16'], help="gradient exchange: 'fp32' bucketed all-reduce (default), 'bf16' all-to-all + fp32 accumulation + all-gather of bf16 payloads (half the bytes per link)")
    ap.add_argument('--comm', default='torch', choices=['torch', 'abi'], help="who issues the collectives: 'torch' = torch.distributed (backend nccl = RCCL; default), 'abi' = the library's own RCCL entry points (rgda_comm_*; the communicator id travels through the process group's store)")
    ap.add_argument('--no-comm-overlap', action='store_true', help='all-reduce the whole gradient after backward instead of bucket by bucket during it')
    ap.add_argument('--eager', action='store_true', help='one Python -> ctypes call per launch instead of the recorded launch plan (regda_amd/plan.py, the default)')
    ap.add_argument('--no-h2d', action='store_true', help='reuse one device-resident batch instead of staging a fresh pinned host batch per step over a copy stream')
    ap.add_argument('--graph', action='store_true', help='replay the step as one captured hipGraph (measured slower than the recorded plan on ROCm 7.2: 22.4 vs 21.0 ms/step)')
    ap.add_argument('--bn-operand', default=None, help="tuning: which BatchNorm units run on their consumer's operand path, e.g. 'stem+bn1+bn2' (model default), 'none'; ':1' appended = wherever served, not only where it pays")
    ap.add_argument('--relu-mask', type=int, default=None, help='tuning: keep ReLU sign bits for units with at least this many channels (model default: all units)')
    ap.add_argument('--wgrad-group-gflop', type=float, default=None, help='tuning: weight gradients are queued and launched in groups of at least this much work (model default: 500)')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--cpu-threads', type=int, default=0, help='threads of the CPU baseline (default: physical cores of one socket within the affinity mask / cgroup quota)')
    ap.add_argument('--phases', action='store_true', help='HIP-event phase marks of one extra step, to stderr')
    ap.add_argument('--align-steps', type=int, default=0, help='also time this many stage-2 ("align", SURVEY 8f.2) '
                    'iterations on the same model and batch; adds "align_step" to the JSON')
    ap.add_argument('--tta-tiles', type=int, default=0, help='also time the teacher harness (SURVEY 8f.1): 8-view TTA '
                    'sliding-window inference of this many 512x512 target tiles; adds "teacher_harness" to the JSON')
    return ap.parse_args()


def flush_c_stdio():
    """Drain the C library's stdout buffer (RCCL's init banner lives there when stdout is a pipe or a file)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def conv_flops_probe(step_fn, park_ms=150.0):
    """Run one step with HIP events (torch.cuda.Event on the launch stream) around every conv launch.
    Returns {kernel instantiation: dict(gflop, ms, launches, tflops, avg_us)}; names match rocprofv3's."""
    import torch
    from regda_amd import ops
    from regda_amd._lib import lib
    L = lib()
    rec = []
    import ctypes
    o_conv, o_wgrad, o_bne, o_bnb = ops.conv2d, ops.conv2d_wgrad, ops.conv2d_bneval, ops.conv2d_bnbwd
    o_wgrad_g, o_bnin, o_conv_g = ops.conv2d_wgrad_grouped, ops.conv2d_bnin, ops.conv2d_grouped

    kname = L.raw('rgda_conv2d_kernel')
    wname = L.raw('rgda_conv2d_wgrad_kernel')

    def timed(kind, variant, w, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, has_stats, groups, launch, extra_bytes=0.0):
        co, taps, ci = w.shape
        M, in_rows = N * Ho * Wo, N * H * W
        # algorithmic HBM bytes of the launch: input rows + weights + output (+ what the fused epilogue reads)
        abytes = 2.0 * in_rows * ci + 2.0 * co * taps * ci + 2.0 * M * co + extra_bytes
        # the instantiation the library's own dispatch picks for this call (rocprofv3's kernel name)
        name = kname(variant, N, H, W, ci, Ho, Wo, co, kh, kw, stride, pad, dil, mode, int(has_stats), groups)
        name = name.decode() if name else 'conv (unserved)'
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        rec.append((name, 2.0 * M * co * taps * ci, e0, e1, (kind, M, co, ci, taps, stride, dil), 1, abytes,
                    2.0 * M * co * taps * ci if taps == 9 else 0.0))

    def conv(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode=0, res=None, stats=None, stat_groups=1, **k):
        M, co = N * Ho * Wo, w.shape[0]
        extra = (2.0 * M * co if res is not None else 0.0) + (M * co / 8.0 if k.get('res_mask') is not None else 0.0)
        timed('dgrad' if mode else 'fwd', 0, w, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, stats is not None, stat_groups,
              lambda: o_conv(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, res, stats, stat_groups, **k), extra)

    def conv_grouped(items):
        # one bracket for the whole list: the library packs the small-tile problems into one launch per eight
        if not items:
            return
        fl = by = 0.0
        for it in items:
            x, w, y, N, H, W, Ho, Wo, kh, kw = it[:10]
            co, taps, ci = w.shape
            fl += 2.0 * N * Ho * Wo * co * taps * ci
            by += 2.0 * N * H * W * ci + 2.0 * co * taps * ci + 2.0 * N * Ho * Wo * co + \
                (2.0 * N * Ho * Wo * co if (len(it) > 14 and it[14] is not None) else 0.0)
        nl = ops.conv2d_grouped_launches(items)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o_conv_g(items)
        e1.record()
        x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil = items[0][:13]
        rec.append(('conv_igemm_grouped_kernel<128, 64, 3, 2, 2, true>', fl, e0, e1,
                    ('grouped x%d' % len(items), N * Ho * Wo, w.shape[0], w.shape[2], kh * kw, stride, dil), nl, by, 0.0))

    def conv_bnin(bnop, x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, res=None, stats=None, stat_groups=1):
        M, co = N * Ho * Wo, w.shape[0]
        timed('fwd<-bn', 3, w, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, 0, True, stat_groups,
              lambda: o_bnin(bnop, x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, res, stats, stat_groups),
              2.0 * M * co if res is not None else 0.0)

    def conv_bne(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, rm, rv, gamma, beta, relu, res=None, **k):
        M, co = N * Ho * Wo, w.shape[0]
        timed('fwd-ev', 1, w, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, 0, False, 1,
              lambda: o_bne(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, rm, rv, gamma, beta, relu, res, **k),
              2.0 * M * co if res is not None else 0.0)

    def conv_bnb(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, res, sums, groups, bn_y, bn_x, *a, **k):
        M, co = N * Ho * Wo, w.shape[0]
        extra = (2.0 * M * co if res is not None else 0.0) + 2.0 * M * co          # residual, consumer's raw conv output
        extra += 2.0 * M * co if bn_y is not None else (M * co / 8.0 if k.get('relu_mask') is not None else 0.0)
        extra += M * co / 8.0 if k.get('res_mask') is not None else 0.0
        timed('dgrad+bn' if mode else 'fwd+bn', 2, w, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, True, groups,
              lambda: o_bnb(x, w, y, N, H, W, Ho, Wo, kh, kw, stride, pad, dil, mode, res, sums, groups, bn_y, bn_x, *a, **k),
              extra)

    def wgrad_kernel_name(item):
        x, dy, dw, N, H, W, Ho, Wo, kh, kw, stride, pad, dil = item
        S, T, Cin = dw.shape
        stacked = kh * kw == 1 and T > 1
        d = ops._WgradDesc()
        d.x, d.dy, d.dw, d.ldx, d.lddy = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), x.stride(0), dy.stride(0)
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Ho, Wo, (S * T if stacked else S)
        d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, stride, pad, dil
        d.lddw, d.co_split = dw.stride(1), (S if stacked else 0)
        name = wname(ctypes.byref(d))
        return name.decode() if name else 'conv_wgrad (unserved)'

    def wgrad_grouped(items):
        # the library buckets a list by kernel instantiation (order kept, 16 layers per launch); hand it one
        # bucket at a time so the events bracket the launches of one instantiation
        buckets = {}
        for it in items:
            buckets.setdefault(wgrad_kernel_name(it), []).append(it)
        for name, its in buckets.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_wgrad_g(its)
            e1.record()
            fl = sum(2.0 * it[3] * it[6] * it[7] * it[2].numel() for it in its)
            by = sum(2.0 * it[3] * it[4] * it[5] * it[2].shape[2] + 2.0 * it[3] * it[6] * it[7] * it[2].shape[0] +
                     4.0 * it[2].numel() for it in its)
            rec.append((name, fl, e0, e1, ('wgrad x%d' % len(its),) + (its[0][3] * its[0][6] * its[0][7],) + tuple(its[0][2].shape[i] for i in (0, 2, 1)) + (its[0][10], its[0][12]),
                        -(-len(its) // 16), by,
                        sum(2.0 * it[3] * it[6] * it[7] * it[2].numel() for it in its if it[8] == 3 and it[9] == 3)))

    def wgrad(*item):
        wgrad_grouped([item])
    ops.conv2d, ops.conv2d_wgrad, ops.conv2d_bneval, ops.conv2d_bnbwd = conv, wgrad, conv_bne, conv_bnb
    ops.conv2d_wgrad_grouped, ops.conv2d_bnin, ops.conv2d_grouped = wgrad_grouped, conv_bnin, conv_grouped
    # The wrappers make the host slower than the GPU for the short kernels, and an event pair then also brackets the
    # host's enqueue time between `e0.record()` and the launch.  Park the stream behind a spin kernel long enough for
    # the host to enqueue the whole step first: every bracket then measures queue-to-queue GPU time only.
    torch.cuda.synchronize()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record(); torch.cuda._sleep(1_000_000); c1.record(); torch.cuda.synchronize()
    cycles_per_ms = 1_000_000 / max(c0.elapsed_time(c1), 1e-3)
    # An event pair also brackets the two event packets themselves (a few microseconds on this runtime): measure that
    # on empty brackets queued behind the same kind of park and take it out of every launch's bracket, so that the
    # per-launch averages are kernel durations -- what `rocprofv3 --kernel-trace --stats` reports (profiles/).
    torch.cuda._sleep(int(5 * cycles_per_ms))
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a0, a1 in empty:
        a0.record(); a1.record()
    torch.cuda.synchronize()
    bracket_ms = sorted(a0.elapsed_time(a1) for a0, a1 in empty)[len(empty) // 2]
    try:
        torch.cuda._sleep(int(park_ms * cycles_per_ms))
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.conv2d, ops.conv2d_wgrad, ops.conv2d_bneval, ops.conv2d_bnbwd = o_conv, o_wgrad, o_bne, o_bnb
        ops.conv2d_wgrad_grouped, ops.conv2d_bnin, ops.conv2d_grouped = o_wgrad_g, o_bnin, o_conv_g
    kern, shapes = {}, {}
    c3 = [0.0, 0.0]                                # FLOPs / ms of the 3x3 convolutions (forward, data and weight gradient)
    for r in rec:
        name, fl, e0, e1, shp, nl, by, fl3 = r     # nl = kernel launches inside the bracket, by = algorithmic bytes
        dt = max(e0.elapsed_time(e1) - bracket_ms, 1e-4)
        c3[0] += fl3; c3[1] += dt * fl3 / max(fl, 1.0)      # a mixed weight-gradient bucket is split by FLOPs
        k = kern.setdefault(name, [0.0, 0.0, 0, 0.0])
        k[0] += fl; k[1] += dt; k[2] += nl; k[3] += by
        q = shapes.setdefault(shp, [0.0, 0.0, 0])
        q[0] += fl; q[1] += dt; q[2] += nl
    if os.environ.get('RGDA_CONV_REPORT'):
        with open(os.environ['RGDA_CONV_REPORT'], 'w') as f:
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                f.write('%-8s M=%-7d Co=%-5d Ci=%-5d taps=%d s=%d d=%d  n=%-3d ms=%8.3f  TF/s=%7.1f\n' % (k + (v[2], v[1], v[0] / 1e9 / max(v[1], 1e-9))))
    out = {k: dict(gflop=v[0] / 1e9, ms=v[1], launches=v[2], tflops=v[0] / 1e9 / max(v[1], 1e-9),
                   avg_us=v[1] / v[2] * 1e3, algorithmic_mb=v[3] / 1e6, gbps=v[3] / 1e6 / max(v[1], 1e-9))
           for k, v in kern.items()}
    out['__bracket_us__'] = bracket_ms * 1e3
    out['__conv3x3__'] = dict(gflop=c3[0] / 1e9, ms=c3[1], tflops=c3[0] / 1e9 / max(c3[1], 1e-9))
    return out


def host_cores():
    """How many threads the CPU baseline may use: the physical cores of ONE socket (/proc/cpuinfo: distinct
    (physical id, core id) pairs of the first package), intersected with this process's affinity mask and capped by
    the cgroup CPU quota (cpu.max / cfs_quota_us).  Returns (threads, dict describing each limit)."""
    info = {'logical_cpus': os.cpu_count() or 1}
    cores_by_pkg, cpu_core = {}, {}
    try:
        cur = {}
        for line in open('/proc/cpuinfo'):
            if ':' in line:
                k, v = (t.strip() for t in line.split(':', 1))
                cur[k] = v
            elif cur:
                if 'processor' in cur:
                    pkg, core = cur.get('physical id', '0'), cur.get('core id', cur['processor'])
                    cores_by_pkg.setdefault(pkg, set()).add(core)
                    cpu_core[int(cur['processor'])] = (pkg, core)
                cur = {}
        if cur and 'processor' in cur:
            pkg, core = cur.get('physical id', '0'), cur.get('core id', cur['processor'])
            cores_by_pkg.setdefault(pkg, set()).add(core)
            cpu_core[int(cur['processor'])] = (pkg, core)
    except OSError:
        pass
    info['sockets'] = max(len(cores_by_pkg), 1)
    first = sorted(cores_by_pkg)[0] if cores_by_pkg else None
    info['physical_cores_per_socket'] = len(cores_by_pkg[first]) if first is not None else info['logical_cpus']
    try:
        aff = os.sched_getaffinity(0)
    except AttributeError:
        aff = set(range(info['logical_cpus']))
    info['affinity_cpus'] = len(aff)
    # physical cores of the first socket this process is allowed on
    allowed = {cpu_core[c] for c in aff if c in cpu_core}
    if allowed:
        pk = sorted({p for p, _ in allowed})[0]
        n = len({c for p, c in allowed if p == pk})
    else:
        n = min(info['physical_cores_per_socket'], len(aff))
    quota = None
    for path, parse_q in (('/sys/fs/cgroup/cpu.max', lambda t: None if t.split()[0] == 'max' else float(t.split()[0]) / float(t.split()[1])),
                          ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', None)):
        try:
            t = open(path).read().strip()
            if parse_q is not None:
                quota = parse_q(t)
            elif int(t) > 0:
                quota = int(t) / float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    info['cgroup_cpu_quota'] = quota
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n), info


def _r(v, nd=4):
    """Floats of the stdout line: 4 significant digits."""
    if isinstance(v, float):
        return float(f'{v:.{nd}g}')
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def compact_line(res, full_path):
    """The ONE stdout line: the bench contract's keys, `roofline` (dominant kernel + the 3x3 / all-convolution figures the
    north star is stated on) and `cpu_baseline`, nothing else; everything `res` holds is in `full_path`."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'pairs_per_sec_per_gpu', 'rccl_ranks', 'plan_replay',
            'host_enqueue_ms_per_step')
    out = {k: res[k] for k in keep if k in res}
    if res.get('with_h2d_staging'):
        out['with_h2d_staging_pairs_per_s'] = res['with_h2d_staging']['pairs_per_s']
    if res.get('comm_exposed_ms'):
        c = res['comm_exposed_ms']
        out['comm_exposed_ms'] = {k: c[k] for k in ('rank0_ms', 'max_over_ranks_ms', 'payload', 'issued_by') if k in c}
    if res.get('replicas'):
        out['replicas_identical'] = res['replicas']['identical']
    rf = res.get('roofline')
    if rf:
        out['roofline'] = {k: rf[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel',
                                              'launches_per_step', 'avg_launch_us', 'algorithmic_mb_per_launch',
                                              'mfma_frac', 'step_executed_mfma_frac')}
        out['roofline']['conv3x3'] = {k: rf['conv3x3'][k] for k in ('achieved', 'frac', 'unit', 'ms_per_step')}
        out['roofline']['all_conv_kernels'] = {'achieved': rf['all_conv_kernels']['achieved'], 'unit': 'TFLOP/s',
                                               'frac': rf['all_conv_kernels']['frac'],
                                               'ms_per_step': rf['all_conv_kernels']['ms_per_step']}
        if rf.get('conv1x1_stream'):
            out['roofline']['conv1x1_stream_hbm_frac'] = rf['conv1x1_stream']['frac']
    cb = res.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'],
                               'sample': f"oracle/step.py (PyTorch CPU fp32 port), b={cb.get('batch', '?')}+{cb.get('batch', '?')}, "
                                         f"{cb['s_per_step']:.2f} s/step on {cb['cores']} threads"}
    for k in ('align_step', 'teacher_harness'):
        if res.get(k):
            out[k + '_ms'] = res[k].get('ms_per_step', res[k].get('ms_per_tile'))
    out['full_record'] = os.path.relpath(full_path, ROOT) if full_path else None
    return _r(out)


def cpu_baseline(args):
    """The CPU oracle (oracle/step.py: the reference's step restated in stock PyTorch fp32) on a bounded
    sample: b = cpu_batch + cpu_batch images, 1 warm-up + timed steps until ~20 s.  Threads = physical cores of one
    socket within the affinity mask / cgroup quota; a short 8-thread probe step is reported next to it so
    the scaling (or the limit that prevents it) can be read off the line."""
    import torch
    from oracle import model as omodel
    from oracle.step import CpuStep
    from regda_amd.synthetic import make_batch
    threads, limits = host_cores()
    if args.cpu_threads:
        threads = args.cpu_threads
    torch.set_num_threads(threads)
    b = args.cpu_batch
    batch = make_batch(b=b, size=args.size, seed=2333, device='cpu')
    sd = omodel.init_state_dict(args.model, 6, seed=0)
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(0))
    # the same work as the GPU leg: with the online EMA teacher (the default) the soft target labels come from the
    # teacher's eval forward on the shadow weights inside the step, and the shadow is updated behind the optimizer
    teacher = not args.no_teacher
    st = CpuStep(sd, protos, resnet_type=args.model, ema_decay=0.999 if teacher else None)
    soft_in = None if teacher else batch['soft_t']
    run = lambda: st.step(batch['images_s'], batch['label_s'], batch['images_t'], soft_in, batch['regs_t'], lr=1e-4)
    t0 = time.time()
    run()
    warm = time.time() - t0
    n, t0 = 0, time.time()
    phases = {}
    while n < 1 or (time.time() - t0 < 15.0 and n < 5):
        ph = {}
        st.step(batch['images_s'], batch['label_s'], batch['images_t'], soft_in, batch['regs_t'], lr=1e-4, phases=ph)
        for k, v in ph.items():
            phases[k] = phases.get(k, 0.0) + v
        n += 1
    dt = (time.time() - t0) / n
    probe = None
    if threads > 8:         # one step at 8 threads (the container probe of SURVEY 8d): does the box scale beyond it?
        torch.set_num_threads(8)
        t1 = time.time()
        run()
        probe = time.time() - t1
        torch.set_num_threads(threads)
    return dict(value=b / dt, unit='pairs/s', cores=threads, threads=threads, kind='port', host=limits, batch=b,
                s_per_step=dt, phases_s={k: v / n for k, v in phases.items()},
                s_per_step_8_threads=probe,
                sample=f'oracle/step.py (stock PyTorch CPU fp32, ' + ('online EMA teacher: eval forward + shadow update inside the step' if teacher else 'offline soft labels') + f'), {args.model}, b={b}+{b} {args.size}x{args.size}, '
                       f'{n} timed step(s) after 1 warm-up ({warm:.1f}s), {dt:.2f} s/step on {threads} threads '
                       f'(= physical cores of one socket within affinity / cgroup quota)')


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-exec this command line under
    `python -m torch.distributed.run --nproc-per-node N` (one process per GPU) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:         # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    gpus_given = args.gpus is not None
    if not gpus_given:          # under a launcher (torchrun ... bench.py) its world size is adopted; alone: one GPU
        args.gpus = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(respawn_under_launcher(args))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:      # only an EXPLICIT --gpus that disagrees with the launcher is an error
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if args.backend == 'gloo':          # launcher self-test on a CPU box: rendezvous, one all-reduce, one JSON line
        dist.init_process_group('gloo')
        t = torch.ones(1) * (dist.get_rank() + 1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({'launcher_selftest': True, 'n_gpus': world, 'ranks': dist.get_world_size(), 'sum': float(t.item())}))
        dist.destroy_process_group()
        return
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get('RGDA_FORCE_DDP'):       # RGDA_FORCE_DDP=1: exercise the RCCL path on one GPU
        # no `device_id=`: binding the process group to the device makes PyTorch initialise the communicator eagerly
        # AND hook the caching allocator's segments into RCCL -- measured on MI355X at world size 1: every kernel of the
        # step slows down, 21.2 -> 24.1 ms/step, before a single collective is issued (scripts/dev/dev_ddp_probe2.py).
        # The lazy form (communicator created by the first collective on the current device) costs nothing.
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                init_method=None if 'MASTER_ADDR' in os.environ else 'tcp://127.0.0.1:29533')
        if dist.get_world_size() != args.gpus or dist.get_backend() != 'nccl':
            sys.exit(f'bench.py: --gpus {args.gpus} but the RCCL process group has {dist.get_world_size()} ranks '
                     f'(backend {dist.get_backend()}): refusing to report a {args.gpus}-GPU number')
    from regda_amd.models.Encoder import Deeplabv2
    from regda_amd.ssl import SSLStep
    from regda_amd.synthetic import make_batch
    from regda_amd.utils.tools import lr_poly, lr_warmup

    torch.manual_seed(2333)
    model = Deeplabv2(dict(backbone=dict(resnet_type=args.model, output_stride=16, pretrained=False),
                           multi_layer=True, cascade=False, use_ppm=True,
                           ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6,
                           is_ins_norm=True))
    # random-init classifiers emit near-uniform probabilities, so no pseudo label would pass the 0.6 threshold and
    # the target loss / its gradient would be identically zero (zero operands also let the chip clock higher):
    # scale the two 6-class classifiers so the teacher is confident on part of the pixels.
    with torch.no_grad():
        for head in ('layer5', 'layer6'):
            model.convs[f'{head}.conv_last.4'].w.mul_(40.0)
    if args.bn_operand is not None:
        units, _, level = args.bn_operand.partition(':')
        model.bn_operand_units = set(units.split('+')) - {'none'}
        if level:
            model.bn_operand_level = int(level)
    if args.relu_mask is not None:
        model.relu_sign_mask = args.relu_mask
    if args.wgrad_group_gflop is not None:
        model.wgrad_group_gflop = args.wgrad_group_gflop
    model.sync_weights()
    if world > 1:       # identical initial weights on every rank
        dist.broadcast(model.flat_p, 0)
        dist.broadcast(model.flat_buf, 0)
        model.sync_weights()
    protos = torch.randn(6, 2048, generator=torch.Generator().manual_seed(0))
    teacher = not args.no_teacher
    comm = None
    if args.comm == 'abi' and dist.is_initialized():
        from regda_amd.ddp import RcclComm
        comm = RcclComm.from_torch_store()
    step = SSLStep(model, protos, ema_decay=0.999 if teacher else None, overlap_wgrad=not args.serial, overlap_comm=not args.no_comm_overlap,
                   grad_payload=args.grad_payload, comm=comm)
    step.measure_comm = world > 1 or bool(os.environ.get('RGDA_FORCE_DDP'))
    batch = make_batch(b=args.batch, size=args.size, seed=2333 + rank, with_soft=not teacher)
    soft = batch.get('soft_t')
    it = [0]
    pf = [None]         # the input prefetcher, created behind the warm-up (below)

    def one():
        i = it[0]
        lr = lr_warmup(1e-2, i, 300) if i < 300 else lr_poly(1e-2, i, 9000, 0.9)   # tools.py:191-207
        it[0] += 1
        if pf[0] is None:
            return step.step(batch['images_s'], batch['label_s'], batch['images_t'], soft, batch['regs_t'], lr)
        b = pf[0].next()
        out = step.step(b['images_s'], b['label_s'], b['images_t'], b.get('soft_t'), b['regs_t'], lr)
        pf[0].release(step.inputs_consumed() if pf[0].single else None)
        return out

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    flush_c_stdio()         # every rank: the communicator exists now, its banner goes out here, not behind the JSON line
    graphed = planned = False
    if not args.graph and not args.eager:      # (the collectives of a multi-GPU step are host actions of the plan)
        try:        # the step is a static launch sequence: replay it below the ABI (one C loop per segment)
            step.record_plan(batch['images_s'], batch['label_s'], batch['images_t'], soft, batch['regs_t'])
            # "inputs resident in HBM" = the recorded step's own input buffers: handing them back makes the replay's input
            # copy a no-op (another device tensor would be copied into them first: 84 MB device -> device per step)
            batch = dict(batch, **{k: v for k, v in step.static_inputs().items() if v is not None})
            soft = batch.get('soft_t')
            it[0] += 1
            one()
            torch.cuda.synchronize()
            planned = True
        except Exception as e:      # stay eager, say so
            print('plan recording failed, running eagerly:', repr(e)[:300], file=sys.stderr)
            step._plan = None
    if world == 1 and args.graph:
        try:        # replay the (static) step as one hipGraph: removes ~20 ms/step of host launch work
            step.capture(batch['images_s'], batch['label_s'], batch['images_t'], soft, batch['regs_t'])
            one()
            torch.cuda.synchronize()
            graphed = True
        except Exception as e:      # stay eager, say so
            print('graph capture failed, running eagerly:', repr(e)[:200], file=sys.stderr)
            step._graph = None
    def timed_steps():
        """EXACTLY args.steps steps between barrier + synchronize on both sides -> (seconds, max over ranks; host seconds; last outputs)."""
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = one()
        th = time.perf_counter() - t0          # host launch work only (the GPU may still be running)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d, th, o

    # `value`: inputs resident in HBM when the timed region starts (the bench contract): the FIRST window of exactly K steps;
    # further windows of the same K steps only report the spread (a 0.4 - 0.8 s window on a freshly leased box moves by +-1 %)
    dt, t_host, out = timed_steps()
    windows = [dt / args.steps * 1e3] + [timed_steps()[0] / args.steps * 1e3 for _ in range(max(0, args.repeats))]
    dt_h2d = None
    if not args.no_h2d:
        # input path (tools/train_ssl_reg.py:200-206 moves every batch to the GPU inside the iteration): two distinct
        # synthetic batches in pinned host memory, copied per step by a copy stream -- straight into the recorded step's
        # static input buffers once the running step has read them, or into double-buffered slots in eager mode
        from regda_amd.utils.prefetch import DevicePrefetcher
        host = [{k: v.cpu() for k, v in batch.items()},
                make_batch(b=args.batch, size=args.size, seed=4666 + rank, with_soft=not teacher, device='cpu')]
        pf[0] = DevicePrefetcher(host, into=step.static_inputs() if planned else None)
        one()
        torch.cuda.synchronize()
        dt_h2d = timed_steps()[0]       # the same K steps with the reference's per-iteration host -> device move
    losses = [float(x.item()) for x in out]
    # exposed communication of the LAST timed step: how long the main stream stood at the join with the gradient exchange
    # (HIP events around reducer.finish); every rank reports its own, rank 0 also the maximum
    comm = None
    if step.comm_events is not None:
        torch.cuda.synchronize()
        mine = step.comm_events[0].elapsed_time(step.comm_events[1])
        print('rank %d: exposed gradient-exchange wait %.3f ms/step, rccl ranks %d, payload %s, %d buckets' % (
            rank, mine, dist.get_world_size() if dist.is_initialized() else 1, args.grad_payload, len(step.reducer.buckets)),
            file=sys.stderr, flush=True)
        worst = mine
        if world > 1:
            t = torch.tensor([mine], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float(t.item())
        comm = {'rank0_ms': mine, 'max_over_ranks_ms': worst, 'payload': args.grad_payload, 'issued_by': args.comm,
                'mb_per_step_per_rank': step.reducer.flat_g.numel() * (4 if args.grad_payload == 'fp32' else 2) / 1e6,
                'what': 'main-stream stall at the join with the bucketed gradient exchange (HIP events around reducer.finish), last timed step'}
        # when each bucket left (its gradients final), relative to the start of backward and to the join -- what overlaps
        if step.bwd_start_event is not None and step.reducer.issue_events:
            per = [{'bucket': i, 'mb': (step.reducer.buckets[i][1] - step.reducer.buckets[i][0]) * 4 / 1e6,
                    'issued_ms_after_backward_start': step.bwd_start_event.elapsed_time(ev),
                    'issued_ms_before_join': ev.elapsed_time(step.comm_events[0])} for i, ev in step.reducer.issue_events]
            comm['buckets'] = per
            print('rank %d buckets: %s' % (rank, ', '.join('#%d %.0f MB @%.2f ms (join -%.2f)' % (
                b['bucket'], b['mb'], b['issued_ms_after_backward_start'], b['issued_ms_before_join']) for b in per)),
                file=sys.stderr, flush=True)
    # ---- every rank must hold the SAME weights after the timed loop (identical initial weights, summed gradients,
    # identical optimizer): a bitwise checksum of the master weights, the EMA shadow and the prototypes is compared over the
    # ranks and the run FAILS if they diverged -- a scaling number from ranks that train different models means nothing
    replicas = None
    if dist.is_initialized():
        torch.cuda.synchronize()
        def bits(t):
            return t.contiguous().view(torch.int32).to(torch.int64).sum()
        parts = [bits(model.flat_p), bits(step.prototypes)] + ([bits(step.teacher.flat_p)] if step.teacher is not None else [])
        lo = torch.stack(parts)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas = {'identical': bool(torch.equal(lo, hi)), 'checksums': [int(v) for v in lo.tolist()],
                    'what': 'int64 sums of the fp32 bit patterns of weights / prototypes / EMA shadow, min == max over ranks'}
        if not replicas['identical']:
            sys.exit(f'bench.py: rank {rank}: the ranks hold DIFFERENT weights after the timed loop '
                     f'(checksums min {lo.tolist()} max {hi.tolist()}): the data-parallel step is broken')
    # host cost of enqueueing ONE step, measured from an idle queue (in the timed loop above the GPU is the bottleneck and
    # the launch queue pushes back on the host, so that loop's host time mostly shows the back-pressure)
    t_iso = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one()
        t_iso.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    if args.phases and rank == 0:
        saved, step._plan = step._plan, None        # the phase marks are recorded by the eager step
        one()                       # keep the GPU queue primed like in the timed loop
        step.marks = []
        one()
        marks, step.marks = step.marks, None
        one()
        torch.cuda.synchronize()
        step._plan = saved
        base = marks[0][1]
        for name, ev in marks:
            print('  %+9.3f ms  %s' % (base.elapsed_time(ev), name), file=sys.stderr)
    pairs = args.batch * world * args.steps
    value = pairs / dt
    gflop_pair = GFLOP_PER_PAIR_STUDENT + (GFLOP_PER_PAIR_TEACHER if teacher else 0.0)
    res = {
        'metric': 'src+tgt 512x512 image-pairs/sec (SSL step)', 'value': value, 'unit': 'pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'ms_per_step_windows': windows, 'ms_per_step_median': sorted(windows)[len(windows) // 2],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
        'data': 'synthetic, inputs resident in HBM',
        'config': {'workload': f'st.regda.2potsdam SSL step, {args.model} DeepLabV2(PPM), batch {args.batch}+{args.batch} '
                               f'{args.size}x{args.size} per GPU, ' + ('online EMA teacher' if teacher else 'offline soft labels'),
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}', 'gflop_per_pair': gflop_pair},
        'pairs_per_sec_per_gpu': value / world,
        # the same steps with a fresh pinned host batch moved to the device per step on a copy stream
        # (tools/train_ssl_reg.py:200-206 does that move inside the iteration); never `value`
        'with_h2d_staging': None if dt_h2d is None else {
            'pairs_per_s': pairs / dt_h2d, 'ms_per_step': dt_h2d / args.steps * 1e3,
            'mb_per_step': pf[0].bytes_per_batch / 1e6},
        'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
        'comm_exposed_ms': comm,
        'replicas': replicas,
        # whole-step MFMA fraction on the REFERENCE's convolution FLOPs (1268 GFLOP/pair with the teacher forward): an
        # "effective" figure -- the step executes fewer (the head conv is re-associated); the executed-FLOP
        # fraction is roofline.step_executed_mfma_frac
        'step_mfma_frac': value / world * gflop_pair / (MFMA_PEAK_TFLOPS * 1e3),
        'step_mfma_frac_basis': 'reference conv FLOPs (BASELINE.md section 2), label path / BN / optimizer time included',
        'loss_source': losses[0], 'loss_target': losses[1], 'hip_graph': graphed, 'plan_replay': planned,
        'host_enqueue_ms_per_step': sorted(t_iso)[1] * 1e3,         # one step enqueued into an idle queue (median of 3)
        'host_loop_ms_per_step': t_host / args.steps * 1e3,         # the timed loop's host side (includes queue back-pressure)
    }
    if rank == 0 and world == 1 and not args.no_roofline:
        step._graph = step._plan = None          # the per-launch HIP-event probe needs the eager path ...
        side, step.wgrad_stream = step.wgrad_stream, None     # ... and one stream, so a launch's events bracket only itself
        kern = conv_flops_probe(one)
        c3 = kern.pop('__conv3x3__')
        bracket_us = kern.pop('__bracket_us__')
        step.wgrad_stream = side
        # the dominant kernel = the conv instantiation with the most GPU time in the step
        dom = max(kern, key=lambda k: kern[k]['ms'])
        d = kern[dom]
        gf = sum(v['gflop'] for v in kern.values())
        ms = sum(v['ms'] for v in kern.values())
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')      # written from the rocprofv3 --pmc passes
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom)
        # which roof bounds it: algorithmic FLOP per algorithmic byte against the machine balance (2500 TF / 8 TB/s)
        intensity = d['gflop'] * 1e3 / max(d['algorithmic_mb'], 1e-9)                # FLOP per byte
        if intensity >= MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBPS:
            roof = {'bound': 'mfma', 'achieved': d['tflops'], 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': d['tflops'] / MFMA_PEAK_TFLOPS}
        else:
            roof = {'bound': 'hbm', 'achieved': d['gbps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': d['gbps'] / HBM_PEAK_GBPS}
        res['roofline'] = {**roof, 'kernel': dom, 'traffic': traffic,
                           'traffic_source': 'profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE per launch from the separate rocprofv3 --pmc passes of `scripts/tune.sh profiles` (a counter pass cannot run inside this process); null if the file is absent',
                           'flop_per_byte': intensity, 'mfma_frac': d['tflops'] / MFMA_PEAK_TFLOPS,
                           'hbm_frac': d['gbps'] / HBM_PEAK_GBPS,
                           'launches_per_step': d['launches'], 'avg_launch_us': d['avg_us'],
                           'event_bracket_us_subtracted': bracket_us,
                           'gflop_per_launch': d['gflop'] / d['launches'],
                           'algorithmic_mb_per_launch': d['algorithmic_mb'] / d['launches'],
                           'all_conv_kernels': {'achieved': gf / ms, 'frac': gf / ms / MFMA_PEAK_TFLOPS,
                                                'gflop_per_step': gf, 'ms_per_step': ms},
                           # conv FLOPs the step actually executes / step time / peak
                           'step_executed_mfma_frac': gf / (dt / args.steps * 1e3) / MFMA_PEAK_TFLOPS,
                           # BASELINE.json's MFMA target is stated on the 3x3 convolutions (forward + both gradients)
                           'conv3x3': {'achieved': c3['tflops'], 'frac': c3['tflops'] / MFMA_PEAK_TFLOPS,
                                       'frac_of_sustained': c3['tflops'] / MFMA_SUSTAINED_TFLOPS,
                                       'unit': 'TFLOP/s', 'gflop_per_step': c3['gflop'], 'ms_per_step': c3['ms']},
                           # the short-K 1x1 convolutions of layers 1 / 2 (conv1x1_stream_kernel, every fused epilogue): HBM-bound
                           'conv1x1_stream': (lambda ks: None if not ks else {
                               'achieved': sum(v['algorithmic_mb'] for v in ks) / max(sum(v['ms'] for v in ks), 1e-9),
                               'frac': sum(v['algorithmic_mb'] for v in ks) / max(sum(v['ms'] for v in ks), 1e-9) / HBM_PEAK_GBPS,
                               'unit': 'GB/s', 'launches_per_step': sum(v['launches'] for v in ks),
                               'ms_per_step': sum(v['ms'] for v in ks)})(
                               [v for k, v in kern.items() if k.startswith('conv1x1_stream_kernel')]),
                           'mfma_sustained_peak': {'value': MFMA_SUSTAINED_TFLOPS, 'unit': 'TFLOP/s',
                                                   'what': 'back-to-back v_mfma_f32_32x32x16_bf16 on N(0,1) operands, no '
                                                           'memory traffic (2440 on zeros); profiles/r02_mfma_ceiling.txt'},
                           'by_kernel': kern}
    if rank == 0 and world == 1 and args.align_steps > 0:
        from regda_amd.align import AlignStep
        ast = AlignStep(model, step.prototypes, overlap_wgrad=not args.serial)
        for _ in range(2):
            ast.step(batch['images_s'], batch['label_s'], batch['images_t'], batch['regs_t'], 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.align_steps):
            oa = ast.step(batch['images_s'], batch['label_s'], batch['images_t'], batch['regs_t'], 1e-3)
        torch.cuda.synchronize()
        dt_a = (time.perf_counter() - t0) / args.align_steps
        res['align_step'] = {'pairs_per_s': args.batch / dt_a, 'ms_per_step': dt_a * 1e3,
                             'loss_seg': float(oa[0].item()), 'loss_align': float(oa[1].item()),
                             'what': 'stage-2 iteration (tools/train_align_reg.py:144-196): no teacher forward, prototype '
                                     'contrastive loss on the features'}
    if rank == 0 and world == 1 and args.tta_tiles > 0:
        from regda_amd.utils.tools import pre_slide
        tm = step.teacher_model() if step.teacher is not None else model       # EMA weights + the student's current BN buffers
        tm.eval()
        tile = batch['images_t'][:1].contiguous()
        with torch.no_grad():
            pre_slide(tm, tile, num_classes=6, tile_size=(args.size, args.size), tta=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.tta_tiles):
                pre_slide(tm, tile, num_classes=6, tile_size=(args.size, args.size), tta=True)
            torch.cuda.synchronize()
        dt_t = (time.perf_counter() - t0) / args.tta_tiles
        res['teacher_harness'] = {'tiles_per_s': 1.0 / dt_t, 'ms_per_tile': dt_t * 1e3, 'views': 8,
                                  'what': f'pre_slide(tta=True) of one {args.size}x{args.size} tile: 8 dihedral views '
                                          'through the eval network as one batch, de-augmented and averaged'}
        model.train()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(args)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # The full record (per-kernel table, windows, host probes, phase splits) goes to a FILE; stdout carries ONE short line
        # (< 1.8 KB: whoever keeps only the tail of stdout still holds every contract key, `roofline` and `cpu_baseline` whole).
        full_path = args.full_json
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, 'w') as f:
                json.dump(res, f)
        except OSError as e:
            print(f'bench.py: could not write {full_path}: {e}', file=sys.stderr)
            full_path = None
        # RCCL prints a version banner through C stdio, which a pipe or file buffers until the process exits -- behind
        # anything Python printed.  Drain it first so that the JSON line is the LAST line of stdout.
        flush_c_stdio()
        print(json.dumps(compact_line(res, full_path)), flush=True)


if __name__ == '__main__':
    main()
