"""The self-training (SSL) inner loop of tools/train_ssl_reg.py:198-241 as ONE fused, sync-free step.

Same composition and hyper-parameters as the reference loop --
    model(src), model(tgt) -> label_refine -> pseudo_selection -> Homogenizer (LRH) -> update_prototype
    -> loss_calc x2 -> backward -> clip_grad_norm_(32) -> SGD(momentum .9, wd 5e-4)
-- but driven directly over the HIP kernel plans (no autograd graph, no host sync, no .item()):
flat gradient buffer, bucketed RCCL all-reduce overlapped with the second backward pass, and one fused
clip + SGD (+ EMA shadow + bf16 weight mirror) kernel.  Optionally the soft target labels come from an
online EMA teacher (regda/utils/ema.py semantics: parameters averaged, BN buffers shared) instead of
the offline `.pt` files of the reference (BASELINE.json north_star; SURVEY.md header table).
"""
import torch

from . import ops
from . import plan
from .ddp import FlatGradReducer, all_reduce_prototype_statistics


class SSLStep:
    def __init__(self, model, prototypes, class_num=6, ignore_label=-1, momentum=0.9, weight_decay=5e-4,
                 max_norm=32.0, cutoff_top=0.8, cutoff_low=0.6, percent=0.5, proto_decay=0.996, refine_temp=2.0,
                 sam_refine=True, refine_label=True, ema_decay=None, max_regions=4096, bucket_elems=12 << 20,
                 process_group=None, overlap_wgrad=True, overlap_comm=True, class_balancer_s=None,
                 class_balancer_t=None, grad_payload='fp32', comm=None):
        self.model = model
        self.C, self.ig = class_num, ignore_label
        self.momentum, self.wd, self.max_norm = momentum, weight_decay, max_norm
        self.top, self.low, self.percent = cutoff_top, cutoff_low, percent
        self.pdecay, self.temp = proto_decay, refine_temp
        self.sam_refine, self.refine_label = sam_refine, refine_label
        self.max_regions = max_regions
        dev = model.device
        self.prototypes = prototypes.to(dev).float().contiguous().clone()
        self.mom = torch.zeros_like(model.flat_p)
        self.lr_dev = torch.zeros(1, device=dev)
        self.gn = torch.zeros(1, device=dev)
        self.gn_ws = torch.zeros(1024, device=dev)
        self.first = True
        self.lrh_ws = None
        self.ema_decay = ema_decay
        self.teacher = None
        if ema_decay is not None:
            self.teacher = model.make_teacher()
        bounds = model.param_boundaries()
        # grad_payload: 'fp32' = bucketed all-reduce of the fp32 gradient; 'bf16' = all-to-all + fp32 accumulation +
        # all-gather of bf16 payloads, half the bytes on every link (regda_amd/ddp.py)
        # comm: a regda_amd.ddp.RcclComm -- the gradient buckets and the prototype statistics then go through the library's
        # own RCCL entry points (rgda_comm_*) instead of torch.distributed
        self.comm = comm
        self.reducer = FlatGradReducer(model.flat_g, bounds, bucket_elems, process_group, payload=grad_payload, comm=comm)
        self.measure_comm = False   # bench: HIP events around the main stream's wait for the gradient exchange
        self.comm_events = None
        self.bwd_start_event = None
        self.wgrad_stream = torch.cuda.Stream(device=dev) if overlap_wgrad else None
        self.source_side = overlap_wgrad    # source half of the label path on the second stream
        # --bcs / --bct of tools/train_ssl_reg.py:54-58,125-158: regda_amd.gast.balance.ClassBalance objects whose
        # frequency EMA re-weights the source / target cross-entropy per class (None = plain CE, the default)
        self.class_balancer_s, self.class_balancer_t = class_balancer_s, class_balancer_t
        self._graph = None
        self._plan = None
        self._proto_ready = None
        self.proto_stats = None     # data-parallel ranks: sums[c][k], cnt[c] of update_prototype (all-reduced per step)
        self.marks = None           # set to [] to collect (name, event) phase marks of the next step (bench --phases)
        self.overlap_comm = overlap_comm
        self.keep_debug = False     # tests: keep the step's target logits / features / refined soft labels (`self.debug`)
        self.debug = None
        self.world = self.reducer.world
        self.group = process_group

    # ------------------------------------------------------------------
    @torch.no_grad()
    def step(self, images_s, label_s, images_t, soft_t, regs_t, lr):
        """One SSL iteration.  Returns device tensors (loss_source, loss_target, grad_norm_sq): nothing here
        synchronises with the host."""
        if self._graph is not None:
            return self._replay(images_s, label_s, images_t, soft_t, regs_t, lr)
        if self._plan is not None:
            return self._replay_plan(images_s, label_s, images_t, soft_t, regs_t, lr)
        ops.set_f32(self.lr_dev, lr)
        with ops.use_stream(torch.cuda.current_stream()):
            return self._step(images_s, label_s, images_t, soft_t, regs_t)

    def capture(self, images_s, label_s, images_t, soft_t, regs_t):
        """Capture one whole step (both streams) into a hipGraph; later `step()` calls replay it.  The step is a
        static launch sequence over fixed shapes, so replay removes the ~20 ms of per-step host launch work.
        Call after at least one eager step (momentum initialisation is a different kernel variant)."""
        assert not self.first, 'run one eager step before capture()'
        if self.world > 1:
            raise RuntimeError('whole-step graphs are single-GPU; the multi-GPU path stays eager')
        if self.class_balancer_s is not None or self.class_balancer_t is not None:
            # ClassBalance.next_class_weight rebinds its frequency tensor and is host arithmetic: a graph would keep the
            # address of the pre-capture tensor and never advance the EMA.  record_plan() re-runs it as a host action.
            raise RuntimeError('class balancing (--bcs / --bct) is host-side state: use record_plan(), not capture()')
        self._static = self._static_inputs(images_s, label_s, images_t, soft_t, regs_t)
        torch.cuda.synchronize()
        m = self.model
        m._hw_ready = m._wt_ready = None          # everything recorded before the synchronize is complete
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            with ops.use_stream(torch.cuda.current_stream()):
                self._out = self._step(*self._static)
                if self.wgrad_stream is not None:   # the weight re-layouts issued on the second stream at the end
                    torch.cuda.current_stream().wait_stream(self.wgrad_stream)     # of the step join the graph
        m._hw_ready = m._wt_ready = None          # (they are events inside the graph now: nothing to wait for outside)
        self._graph = g

    @staticmethod
    def _static_inputs(images_s, label_s, images_t, soft_t, regs_t):
        """Static input buffers of a recorded / captured step, in the dtype and layout the step's kernels read (the
        conversions `_step` would otherwise do with torch ops at record time only): fp32 contiguous images and soft
        labels, int64 contiguous labels and region maps.  `copy_` into them converts whatever the caller passes."""
        conv = lambda t, dt: None if t is None else t.detach().to(dt).contiguous().clone()
        return [conv(images_s, torch.float32), conv(label_s, torch.int64), conv(images_t, torch.float32),
                conv(soft_t, torch.float32), conv(regs_t, torch.int64)]

    def teacher_model(self):
        """The EMA teacher, ready to be evaluated or saved: shadow weights as they stand, BatchNorm buffers adopted from
        the student NOW (the teacher keeps a snapshot, not an alias: with offline soft labels nothing else refreshes
        it), bf16 mirror current, eval mode."""
        t = self.teacher
        if t is None:
            return None
        t.adopt_buffers(self.model)
        t.refresh_from_master(mirror_is_fresh=(not self.first) and t.flat_p._version == t._synced_version)
        t.eval()
        return t

    def record_plan(self, images_s, label_s, images_t, soft_t, regs_t, lr=None):
        """Record one whole step (all streams) as a launch plan (regda_amd/plan.py): its ~750 entry-point calls become
        rows of a table that `rgda_plan_run` walks in C, the torch ops / stream waits between them stay host actions.
        Later `step()` calls replay the plan on the recorded buffers (a private memory pool): same kernels, same
        streams, same results as the eager step, a fraction of the host time.  Call after at least one eager step
        (the first step initialises the momentum with another kernel variant).  Inputs are copied into static
        buffers at every replay; the returned tensors (losses, `last_hard`, ...) are overwritten by the next step."""
        assert not self.first, 'run one eager step before record_plan()'
        assert self._graph is None and self._plan is None
        self._static = self._static_inputs(images_s, label_s, images_t, soft_t, regs_t)
        if lr is not None:          # recording IS a real step: run it with this learning rate (default: the last one)
            ops.set_f32(self.lr_dev, lr)
        torch.cuda.synchronize()
        self._plan_stream = torch.cuda.current_stream()
        p = plan.Plan()

        def run():
            with ops.use_stream(torch.cuda.current_stream()):
                return self._step(*self._static)
        self._out = p.record(run)
        self._plan = p
        return p.stats()

    def release_plan(self):
        self._plan = None

    def static_inputs(self):
        """The recorded step's input buffers {name: device tensor}: a loader may write the next batch straight into
        them once `inputs_consumed()` of the running step has passed (regda_amd/utils/prefetch.py)."""
        assert self._plan is not None or self._graph is not None
        names = ('images_s', 'label_s', 'images_t', 'soft_t', 'regs_t')
        return dict(zip(names, self._static))

    def inputs_consumed(self):
        """Event of the most recently enqueued step: recorded behind the last kernel that reads the step's inputs."""
        return self._inputs_done.ev

    def _replay_plan(self, images_s, label_s, images_t, soft_t, regs_t, lr):
        # the table rows carry the stream handles of the recording: the input copies and the learning-rate fill must
        # be ordered on that same stream, whatever stream the caller has made current
        rec = self._plan_stream
        cur = torch.cuda.current_stream()
        if cur != rec:
            rec.wait_stream(cur)
        with ops.use_stream(rec):       # host-action entry points (class weights, ...) land on the recorded stream too
            for dst, src in zip(self._static, (images_s, label_s, images_t, soft_t, regs_t)):
                assert (dst is None) == (src is None), 'a recorded step keeps its input signature (soft labels or not)'
                if dst is not None and src is not dst:
                    dst.copy_(src, non_blocking=True)
            ops.set_f32(self.lr_dev, lr)
            m = self.model
            if not m.training:
                m.train()
            if m.flat_p._version != m._synced_version:      # weights were changed from outside (load_state_dict, ...)
                m.sync_weights()
            self._plan.replay()
        if cur != rec:
            cur.wait_stream(rec)
        return self._out

    def _replay(self, images_s, label_s, images_t, soft_t, regs_t, lr):
        for dst, src in zip(self._static, (images_s, label_s, images_t, soft_t, regs_t)):
            if dst is not None and src is not dst:
                dst.copy_(src, non_blocking=True)
        ops.set_f32(self.lr_dev, lr)
        self._graph.replay()
        return self._out

    def _class_weights(self, balancer, label):
        """Per-class CE weights of the two heads [2, C] (or None): loss_calc(multi=True) calls the loss once per head
        and every call EMA-updates the balancer (regda/gast/balance.py:27-35), so the heads see consecutive states.
        A host action of a recorded step: the counts and the 6-element arithmetic are re-run at every replay."""
        if balancer is None:
            return None
        cw = torch.empty(2, self.C, device=label.device)

        def update():
            for hd in range(2):
                cw[hd].copy_(balancer.next_class_weight(label))
        plan.host(update)
        return cw

    def _mark(self, name, stream=None):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self.marks.append((name, ev))

    def _step(self, images_s, label_s, images_t, soft_t, regs_t):
        m = self.model
        if not m.training:
            m.train()
        self._mark('step start')
        m._maybe_sync()
        # source and target batch go through the network TOGETHER (twice the GEMM rows per launch), as two
        # BatchNorm groups: statistics, running-stat updates and gradients stay per domain like the
        # reference's two separate forward calls (train_ssl_reg.py:210-212)
        nb = images_s.shape[0]
        T = m.new_tape(groups=2)
        main = torch.cuda.current_stream()
        online = soft_t is None
        teacher_on_side = online and self.wgrad_stream is not None
        if online:
            # the teacher sees the BatchNorm running statistics as they stand at the START of the step (a snapshot on
            # the main stream, before the student's forward rewrites them) -- the same view whether its forward then
            # runs next to the student's on the second stream or after it
            self.teacher.adopt_buffers(m)
        if teacher_on_side:
            # the EMA teacher's forward is independent of the student's: it runs on the second stream, next to it
            plan.wait_stream(self.wgrad_stream, main)
            with ops.use_stream(self.wgrad_stream):
                soft_t = self.teacher_probs(images_t, snapshot=False)
                self._mark('teacher forward done (side)', self.wgrad_stream)
                # the gradient buffer is cleared here, behind the teacher and next to the student's forward (nothing
                # writes it before the main stream has joined this one), instead of ahead of the student's first kernel
                ops.fill_zero(m.flat_g)
        else:
            ops.fill_zero(m.flat_g)
        x1, x2, feat = m._forward_plan([images_s.contiguous().float(), images_t.contiguous().float()], T)
        self._mark('student forward done')
        s1, t1, s2, t2 = x1[:nb], x1[nb:], x2[:nb], x2[nb:]
        # d(loss) / d(logits) of both heads, source rows then target rows: the two loss kernels write their halves in place
        g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
        feat_s, feat_t = feat[:nb], feat[nb:]
        if teacher_on_side:
            plan.wait_stream(main, self.wgrad_stream)
        elif online:
            soft_t = self.teacher_probs(images_t, snapshot=False)
        self.last_soft_t = soft_t
        self._mark('joined teacher')
        # ---- label path (a5-a8)
        def wait_prototypes():           # the previous step's cross-rank average of the prototypes (second stream)
            if self._proto_ready is not None:
                main.wait_event(self._proto_ready)
                self._proto_ready = None
        plan.host(wait_prototypes)
        # the source half of the label path (source loss, prototype update) is independent of the target chain
        # (refine -> select -> LRH -> target loss): with a second stream it runs there, next to the target chain
        side = self.wgrad_stream if self.source_side else None
        if side is not None:
            plan.wait_event(side, plan.record_event(main))
            with ops.use_stream(side):
                loss_s, _, _ = ops.upsample_ce(s1, s2, label_s, self.ig,
                                               self._class_weights(self.class_balancer_s, label_s), True, g1[:nb], g2[:nb])
        # pseudo_selection + LRH in ONE pass over the refined soft labels (rgda_pseudo_lrh: the selected label is never
        # written as an int64 tensor and read back) where the chain is the default one; tests that look at the selected
        # labels (keep_debug) and the other configurations take the two calls
        regs = (regs_t.squeeze(1) if regs_t.dim() == 4 else regs_t) if self.sam_refine else None
        fuse_lrh = (self.refine_label and self.sam_refine and not self.keep_debug and self.C == 6 and self.max_regions <= 65535
                    and (soft_t.shape[-1] * soft_t.shape[-2]) % 4 == 0)
        hard = None
        if self.refine_label:
            soft, cm = ops.label_refine(feat_t, self.prototypes, t1, t2, soft_t, self.temp, return_ws=True)
            if not fuse_lrh:
                hard = ops.pseudo_select(soft, self.top, self.low, self.ig, classmax_ws=cm, check=False)
        else:
            soft = soft_t
            hard = ops.pseudo_select(soft_t, self.top, self.low, self.ig, check=False)
        exchange_protos = self.reducer.active
        if side is not None:
            # update_prototype rewrites the prototypes label_refine has just read
            plan.wait_event(side, plan.record_event(main))
            with ops.use_stream(side):
                self._proto_local(feat_s, label_s, exchange_protos)
            source_done = plan.record_event(side)
        if self.keep_debug:
            self.debug = dict(t1=t1, t2=t2, s1=s1, s2=s2, feat_t=feat_t, feat_s=feat_s, soft_in=soft_t, soft=soft,
                              hard_selected=hard)
        if self.sam_refine:
            self._lrh_flag_off = (regs.shape[0] * self.max_regions * (self.C + 1)) * 4
            if fuse_lrh:
                hard, self.lrh_ws = ops.pseudo_lrh(soft, cm, regs, self.top, self.low, self.percent, self.C, self.ig,
                                                   self.max_regions, ws=self.lrh_ws)
            else:
                need = self._lrh_flag_off + 16
                if self.lrh_ws is None or self.lrh_ws.numel() < need:
                    self.lrh_ws = torch.empty(need, dtype=torch.uint8, device=m.device)
                hard = ops.lrh(hard, regs.contiguous(), self.percent, self.C, self.ig, self.max_regions, check=False,
                               ws=self.lrh_ws)
        if side is None:
            self._proto_local(feat_s, label_s, exchange_protos)
        if exchange_protos:
            # data-parallel ranks: the per-class feature sums and pixel counts of the rank's source batch are
            # all-reduced (sum) and every rank applies the same totals -- the prototypes of the concatenated GLOBAL batch
            # (alignment.py:300-327: sum feat 1[c] / (n_c + eps), old prototype kept iff the global n_c < 1), identical bits on
            # every rank, the reference exactly at world 1.  Nothing of THIS step reads the prototypes any more (label_refine
            # is done), so the 48 KB all-reduce -- pure latency -- and the apply run on the second stream next to backward;
            # the next step's label path waits for them
            pside = self.wgrad_stream if self.wgrad_stream is not None else main
            if pside is not main and side is None:
                plan.wait_event(pside, plan.record_event(main))
            def exchange_statistics():
                all_reduce_prototype_statistics(self.proto_stats, self.C, self.prototypes.shape[1], self.group, self.comm,
                                                world=self.world)
            with ops.use_stream(pside):
                plan.host(exchange_statistics)
                ops.proto_apply(self.prototypes, self.proto_stats, self.pdecay)
                plan.host(lambda: setattr(self, '_proto_ready', pside.record_event() if pside is not main else None))
        # ---- losses + d(loss)/d(logits)
        if side is None:
            loss_s, _, _ = ops.upsample_ce(s1, s2, label_s, self.ig,
                                           self._class_weights(self.class_balancer_s, label_s), True, g1[:nb], g2[:nb])
        loss_t, _, _ = ops.upsample_ce(t1, t2, hard, self.ig, self._class_weights(self.class_balancer_t, hard), True,
                                       g1[nb:], g2[nb:])
        if side is not None:
            plan.wait_event(main, source_done)       # source loss, its logit gradients, the new prototypes
        self._mark('label path + losses done')
        # from here on the step no longer reads its input tensors (images: stem im2col of student and teacher; labels,
        # soft labels and region maps: the label path and the losses): an input prefetcher may overwrite them
        self._inputs_done = plan.record_event(main)
        self._backward_and_update(T, main, g1, g2)
        self.last_hard = hard
        return loss_s, loss_t, self.gn

    def _proto_local(self, feat_s, label_s, exchange):
        """update_prototype on this rank's source batch: the whole update (one rank), or its sufficient statistics only
        (data-parallel ranks: `_step` all-reduces them and applies the totals)."""
        if exchange:
            self.proto_stats, _ = ops.proto_stats(feat_s, label_s, 16, self.ig, 0.75, self.C, stats=self.proto_stats)
        else:
            ops.proto_update(feat_s, label_s, self.prototypes, 16, self.ig, 0.75, self.pdecay)

    def _backward_and_update(self, T, main, g1, g2, gfeat=None):
        """Backward (both domains in one pass; all-reduce buckets are released as it moves down the net), then clip +
        SGD (+ EMA) in one pass over the flat buffers."""
        m = self.model
        plan.host(self.reducer.reset)
        if self.measure_comm and self.reducer.active:
            def mark_backward_start():
                self.reducer.measure = True
                self.bwd_start_event = torch.cuda.Event(enable_timing=True)
                self.bwd_start_event.record()
            plan.host(mark_backward_start)
        T['wgrad_stream'] = self.wgrad_stream
        T['main_stream'] = main
        T['mark'] = self._mark if self.marks is not None else None

        def progress(offset):
            if (self.world > 1 or self.reducer.force) and self.overlap_comm:
                if self.wgrad_stream is not None:
                    # a bucket needs the weight gradients (second stream) AND the BN gradients (main stream) of its
                    # layers: issue it from the second stream once that has caught up with this point of the main
                    # stream, so the critical path never waits for communication
                    plan.wait_event(self.wgrad_stream, plan.record_event(main))
                    with ops.use_stream(self.wgrad_stream):
                        plan.host(lambda: self.reducer.ready_down_to(offset))
                else:
                    plan.host(lambda: self.reducer.ready_down_to(offset))
        m._backward_plan(T, g1, g2, on_progress=progress, gfeat=gfeat)
        self._mark('backward done (streams joined)')
        def finish_exchange():
            # everything the main stream still has to wait for here is EXPOSED communication (the buckets were issued
            # while backward ran); with measure_comm the stall is bracketed by events (bench.py: comm_exposed_ms)
            if self.measure_comm and self.reducer.active:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.reducer.finish()
                e1.record()
                self.comm_events = (e0, e1)
            else:
                self.reducer.finish()
        plan.host(finish_exchange)
        # ---- clip + SGD (+ EMA) in one pass over the flat buffers
        ops.sumsq(m.flat_g, self.gn, self.gn_ws)
        shadow = self.teacher.flat_p if self.teacher is not None else None
        ops.sgd_step(m.flat_p, m.flat_g, self.mom, shadow, m.flat_pb, self.gn, self.lr_dev, self.momentum, self.wd,
                     self.max_norm, self.reducer.gscale, self.ema_decay if shadow is not None else 0.0, self.first,
                     shadow_bf16=self.teacher.flat_pb if shadow is not None else None)   # teacher mirror stays fresh
        self.first = False
        m.sync_derived_weights(self.wgrad_stream)
        m._synced_version = m.flat_p._version
        self._mark('optimizer + weight mirrors done')

    @torch.no_grad()
    def teacher_probs(self, images_t, snapshot=True):
        """Eval-mode forward of the EMA teacher (Encoder.py:152-155 on the shadow weights; BatchNorm buffers = the
        student's, copied now unless the caller already took the snapshot)."""
        t = self.teacher
        if snapshot:
            t.adopt_buffers(self.model)
        # make_teacher() and every sgd_step keep flat_pb current; a write to the shadow from OUTSIDE the step (a resume
        # that copies into teacher.flat_p or its parameter views) bumps the tensor's version and forces a re-cast
        t.refresh_from_master(mirror_is_fresh=(t.flat_p._version == t._synced_version))
        t.eval()
        x1, x2, _ = t._forward_plan(images_t.contiguous().float(), None)
        return ops.teacher_probs(x1, x2, tuple(images_t.shape[-2:]))

    def lrh_flag(self):
        """Deferred check of the LRH kernel's flag word (host sync): region id / label range errors."""
        if self.lrh_ws is None:
            return 0
        o = self._lrh_flag_off
        return int(self.lrh_ws[o:o + 4].view(torch.int32).item())
