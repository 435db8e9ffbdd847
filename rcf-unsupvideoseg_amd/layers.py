"""Layer objects of the HIP training path: a small reverse-mode tape over NHWC activations.

torch.nn.Module is used as the parameter container (so state-dict keys equal the reference's,
SURVEY.md Appendix B, and `main.py`'s optimizer / checkpoint code sees ordinary Parameters), but no
torch autograd graph is built: every `fwd` launches HIP kernels (ops.py) and pushes one closure on
the tape that launches the matching backward kernels.  Gradients w.r.t. activations live in
`Act.grad` and follow an explicit protocol -- the first producer writes (beta 0), later producers
accumulate in the kernel epilogue (beta 1) -- so residual joins cost no extra pass.  Parameter
gradients always accumulate into `param.grad` (zeroed once per step by the trainer).
"""
import torch
import torch.nn as nn

from . import ops


# Storage type of the activations (and their gradients) between layers: torch.float32, or torch.bfloat16 for the
# mixed-precision step (BASELINE configs[2]: bf16 forward / fp32 gradients -- what the reference's AMP configs do under
# torch autocast, configs/rcf_stv2/rcf_stage1.yaml:57-60).  In bf16 mode the loss tail stays fp32: every conv runs on the
# bf16-operand kernels (csrc/igemm_bf16.hip) with fp32 accumulation, the heads' final 1x1 convs write fp32 logits,
# parameter gradients are fp32.  The type is a property of the FORWARD PASS -- `Tape.act_dtype`, set by RCFModel from its
# `precision` / the ambient autocast state -- not of the process: two models of different precision coexist.


def _round_up(n, m):
    return (n + m - 1) // m * m


class Act:
    """An NHWC activation [N,H,W,C] plus its (lazily created) gradient buffer."""
    __slots__ = ("t", "grad", "needs_grad", "stats", "bn", "amax", "grad_amax", "planes", "split", "accepts_plane_grad",
                 "grad_is_planes", "stats_global", "bn_ctx", "first_reader", "grad_sums2", "pending_add", "addend_ok",
                 "relu_out", "grad_masked", "grad_colsum", "bn_in", "relu_bits", "res_norm", "res_ctx", "res_done")

    def __init__(self, t, needs_grad=True):
        self.t, self.grad, self.needs_grad = t, None, needs_grad
        # fp16 pair planes (include/rcf_hip.h RCF_CONV_X_PLANES): `planes` = an fp32-typed buffer of t's shape holding t's values
        # pre-split for the conv kernels ([pixel][h | m], scale from the bound in `amax`), written by the batch norm that produced
        # t; `split` = there is no fp32 copy (t IS planes).  `accepts_plane_grad`: set by the conv that produced t when its
        # backward can take t's gradient in that format -- the batch norm's backward then writes it so (`grad_is_planes`).
        self.planes, self.split, self.accepts_plane_grad, self.grad_is_planes = None, False, False, False
        self.stats_global = False    # `stats` were already all-reduced over the data-parallel ranks (DistCtx.allreduce_sum_many)
        self.stats = None            # fp64 [2C] column sums | sums of squares, when the producing conv computed them
        self.bn = None               # (mean, invstd, count) when the producing conv's reduction also finalized the norm
        self.amax = None             # int32 [1]: raw bits of max |t| (operand range of the fp16-pair conv kernels)
        self.grad_amax = None        # the same for `grad`, when its only producer computed it
        # the batch norm + ReLU that produced t leaves (its input, sign bits, mean, invstd) here; the conv that read t FIRST in
        # the forward pass writes t's gradient LAST in the backward pass and can then deliver the norm's backward sums from its
        # data gradient's epilogue (`grad_sums2`; any later writer of `grad` -- there should be none -- voids them in grad_slot)
        self.bn_ctx = self.first_reader = self.grad_sums2 = None
        # deferred residual gradient (SCHED.defer_residual): a join relu(bn3 + identity) leaves (its output gradient, its sign bits) here
        # instead of writing the identity's gradient; the conv that read t first (`addend_ok`: its data gradient can take a masked
        # addend) adds it in its epilogue.  Anybody else who wants `grad` gets it materialised (relu_mask_copy).
        self.pending_add, self.addend_ok = None, False
        # folded conv + batch norm + ReLU (conv_bn_fold): `relu_out` = t is the output of a ReLU whose backward wants grad under
        # its own mask (t > 0) plus that gradient's column sums; the LAST writer of `grad` (the conv that read t first) may apply
        # the mask in its epilogue and leave `grad_masked` / `grad_colsum` -- any writer after that would add unmasked terms
        self.relu_out, self.grad_masked, self.grad_colsum = False, False, None
        self.relu_bits = None        # ... the ReLU's sign bits in the producing tile's order (ops.conv2d_fwd_affine_bf16 want_bits)
        # SyncBN: the plain norm (no ReLU) that produced t leaves (its input, mean, invstd) here; a join that takes t as its
        # residual computes that norm's backward sums beside its own -- both depend only on the join's masked output gradient --
        # and sends them in ONE exchange (`grad_sums2` then holds (global sums, local sums) and the norm skips pass and exchange)
        self.bn_in = None
        # a plain norm applied LAZILY (BatchNorm2d.fwd lazy=True): t is the norm's INPUT and (mean, invstd, gamma, beta) is what
        # the join that takes t as its residual normalises it by (ops.bn_apply res_norm) -- the norm's own apply pass never runs
        self.res_norm = None
        # ... (the norm module, its input Act, its row count) for the join that runs that norm's backward beside its own
        # (SCHED.merge_downsample_bwd: one reduction + one apply pass for both), and the flag by which it tells the norm's own closure
        self.res_ctx, self.res_done = None, False

    def range(self):
        """max |t| as a device scalar, computed once per activation (every conv reading it shares the value)"""
        if self.amax is None:
            self.amax = ops.absmax(self.t)
        return self.amax

    def _materialize_pending(self):
        dy, mask = self.pending_add
        self.pending_add = None
        if self.grad is None:
            self.grad = ops.relu_mask_copy(dy, mask)
        else:
            ops.relu_mask_copy(dy, mask, out=self.grad, beta=1)
        self.grad_amax = self.grad_sums2 = None

    def grad_slot(self, takes_addend=False):
        """(tensor, beta) for the next gradient producer.  takes_addend: the producer adds `pending_add` itself (it must take it
        with `take_pending` when this returns beta 0)."""
        if self.pending_add is not None and not (takes_addend and self.grad is None):
            self._materialize_pending()
        if self.grad_masked:
            raise ops._lib.RcfHipError("a gradient producer after the one that applied the ReLU mask (Act.grad_masked): the "
                                       "first reader of this activation in the forward pass is not the last writer of its gradient")
        if self.grad is None:
            self.grad = torch.empty(tuple(self.t.shape), dtype=self.t.dtype, device=self.t.device)
            return self.grad, 0
        self.grad_amax = None        # a second producer accumulates: the first one's range no longer bounds the sum
        self.grad_sums2 = None       # ... nor are sums taken over an earlier state of the gradient the sums of the final one
        self.grad_colsum = None
        return self.grad, 1

    def take_pending(self):
        p, self.pending_add = self.pending_add, None
        return p

    def take_grad(self):
        if self.pending_add is not None:
            self._materialize_pending()
        g, self.grad = self.grad, None
        self.grad_is_planes = False
        self.grad_masked, self.grad_colsum = False, None
        return g

    def take_grad_range(self):
        """call before take_grad: max |grad| as a device scalar (from the producer when it left one)"""
        if self.pending_add is not None:
            self._materialize_pending()
        r, self.grad_amax = self.grad_amax, None
        return r if r is not None else ops.absmax(self.grad)


# How the step is scheduled (second stream, fused batch-norm forms, pair planes, the folded conv + norm, ...): ONE documented
# object, rcf_amd.config.SCHED (class config.Schedule), read at call time.  Tools and tests set its fields explicitly.
from .config import SCHED  # noqa: E402

# test hook (tests/test_model_gpu.py::test_train_step_all_grads_at_fixed_relu_pattern): a list that receives, in forward
# order, the boolean map `output > 0` of every batch norm + ReLU -- the activation pattern of this evaluation
RELU_TRACE = None
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device.index)
    if s is None:
        # SCHED.side_priority: -1 = high priority for the second stream (experiment knob; default: the same priority)
        prio = int(SCHED.side_priority)
        s = _side_streams[device.index] = torch.cuda.Stream(device=device, priority=prio)
    return s


def join_side_stream(device=None):
    for idx, s in _side_streams.items():
        if device is None or device.index == idx:
            torch.cuda.current_stream(s.device).wait_stream(s)


class Tape:
    def __init__(self, enabled=True, on_mark=None, act_dtype=torch.float32):
        assert act_dtype in (torch.float32, torch.bfloat16, torch.float16)
        self.ops, self.enabled, self.on_mark, self.act_dtype = [], enabled, on_mark, act_dtype

    def push(self, fn):
        if self.enabled:
            self.ops.append(fn)

    def mark(self, name):
        """Pushed BEFORE a group of layers runs forward, so it fires right AFTER their backward: every parameter
        gradient of the group is final (the trainer starts that chunk's gradient all-reduce there)."""
        if self.enabled and self.on_mark is not None:
            cb = self.on_mark

            def fire():
                join_side_stream()                       # the group's weight gradients are complete
                cb(name)
            self.ops.append(fire)

    def backward(self):
        with ops.half_storage(self.act_dtype):               # the library build whose 16-bit type this pass's activations have
            while self.ops:
                self.ops.pop()()
            join_side_stream()


class DistCtx:
    """Data-parallel context for SyncBN statistics (torch.distributed over RCCL, or gloo in tests).
    One all-reduce per batch norm and direction: forward the fp64 vector [sum x | sum x^2] (2C doubles), backward
    [sum g | sum g xhat] -- never two.  `count` / `profile` let a test or a dry run see how many collectives a step issues
    and what one costs on the stream it runs on (tests/test_dist_gpu.py)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.group = group
        self.world = dist.get_world_size(group) if self.on else 1
        self.count, self.bytes = 0, 0
        self.profile = False
        self.events = []                 # profile mode: (start, end) event pairs around each collective

    def allreduce_sum(self, t):
        if self.on:
            import torch.distributed as dist
            self.count += 1
            self.bytes += t.numel() * t.element_size()
            if self.profile:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                e1.record()
                self.events.append((e0, e1))
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_sum_many(self, ts):
        """ONE collective for several statistics vectors whose producers do not depend on each other (a bottleneck's conv1 and
        downsample conv read the same input): the latency of a SyncBN exchange is per collective, not per byte"""
        ts = [t for t in ts if t is not None]
        if not self.on or not ts:
            return
        if len(ts) == 1:
            self.allreduce_sum(ts[0])
            return
        flat = torch.cat([t.reshape(-1) for t in ts])
        self.allreduce_sum(flat)
        o = 0
        for t in ts:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()

    def profile_ms(self):
        """profile mode: per-collective device time between the events around it (ms), after a synchronize"""
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in self.events]


def _param_grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    return p.grad


class Conv2d(nn.Module):
    """Bias-free (or biased) square conv; weight kept in channels_last memory ([Cout][R][S][Cin]).
    `cin_pad`: the kernels need Cin % 4 == 0; 3-/2-channel inputs arrive zero-padded to 4 and the
    [Cout,Cin,R,S] parameter is re-packed to [Cout][R][S][4] on the fly (state-dict shape unchanged)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, bias=False, act=0, slope=0.0):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.stride, self.padding, self.dilation, self.act, self.slope = stride, padding, dilation, act, slope
        self.cin_pad = (cin + 3) // 4 * 4
        self.cout_pad = (cout + 3) // 4 * 4
        w = torch.empty(cout, cin, k, k)
        if self.cin_pad == cin:
            w = w.contiguous(memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_normal_(self.weight, a=0, mode="fan_out", nonlinearity="relu")   # mmcv kaiming_init

    def _derived(self, kind, make):
        """an operand derived from self.weight (its range, its fp16 planes, its bf16 copies), made once per weight update
        instead of once per launch: valid while the weight's storage, torch version and the library's weight epoch stand"""
        if not SCHED.cache_weight_operands:
            return make()
        if kind.startswith("bf16") and ops.half() == torch.float16:
            kind = "f16" + kind[4:]                      # the 16-bit copies of the fp16 build are another operand
        key = ops.weight_key(self.weight)
        cache = self.__dict__.setdefault("_wcache", {})
        hit = cache.get(kind)
        if hit is not None and hit[0] == key:
            if ops.DEBUG_WEIGHT_CACHE and hit[2] != ops.weight_checksum(self.weight):
                raise ops._lib.RcfHipError(f"stale cached weight operand '{kind}': the weight was written through .data / a raw "
                                           "pointer without rcf_amd.ops.weights_changed()")
            return hit[1]
        v = make()
        cache[kind] = (key, v, ops.weight_checksum(self.weight) if ops.DEBUG_WEIGHT_CACHE else None)
        return v

    def planes_ok(self, act_dtype=torch.float32):
        """can this conv take its input AND its output gradient as fp16 pair planes (forward, data gradient and weight gradient
        on the LDS-DMA kernels)?  Channel counts: 16-channel K-steps inside one tap, 64-channel groups of the weight gradient.
        act_dtype: the forward pass's activation type (Tape.act_dtype): planes exist in the fp32 step only."""
        return (SCHED.planes and SCHED.fp16_pairs and act_dtype == torch.float32 and self.bias is None and not self.act
                and self.cin % 64 == 0 and self.cout % 16 == 0 and self.cout >= 64      # (the plane weight gradient's tiles: >= 64 rows)
                and self.cin_pad == self.cin and self.cout_pad == self.cout)

    def _packed_weight(self, cout_mult=4):
        """weight as the kernels want it: [Cout_pad][R][S][Cin_pad] (channels_last view), zero padded
        (Cout_pad = Cout rounded up to `cout_mult`)."""
        cp = _round_up(self.cout, cout_mult)
        if self.cin_pad == self.cin and cp == self.cout:
            return self.weight
        if self.cin_pad != self.cin:
            # [Cout,Cin,R,S] contiguous is an "NCHW" tensor with N=Cout: the layout kernel pads C to 4
            assert cp == self.cout
            w = ops.nchw_to_nhwc(self.weight.detach().contiguous(), self.cin_pad)      # [Cout,R,S,4]
            return w.permute(0, 3, 1, 2)
        # Cout not a multiple (e.g. 3 segments): rows are contiguous, so padding = prefix copy + zero rows
        w = torch.zeros((cp, self.k, self.k, self.cin), dtype=torch.float32, device=self.weight.device)
        n = self.weight.numel()
        ops.copy2d(self.weight.detach().permute(0, 2, 3, 1), n, w, n, 1, n)
        return w.permute(0, 3, 1, 2)

    def _packed_bias(self):
        if self.bias is None or self.cout_pad == self.cout:
            return self.bias
        b = torch.zeros(self.cout_pad, dtype=torch.float32, device=self.bias.device)
        b[:self.cout].copy_(self.bias.detach())
        return b

    def _fwd_bf16_padded_cin(self, x, tape, out, stats):
        """the stem (3 input channels) on the bf16 kernels: the image arrives as bf16 NHWC zero-padded to 8 channels
        (`BF16_STEM`), the [Cout,Cin,R,S] parameter is re-packed to [Cout][R][S][8] once per weight update.  No data
        gradient (the input is the image)."""
        c8 = _round_up(self.cin, 8)
        if x.t.shape[3] != c8 or x.needs_grad or self.bias is not None or out is not None or self.cout % 8:
            raise ops._lib.RcfHipError("bf16 conv with Cin % 8 != 0: bias-free stem on an 8-channel-padded input only")
        wpad = self._derived("w_cin8", lambda: ops.nchw_to_nhwc(self.weight.detach().contiguous(), c8).permute(0, 3, 1, 2))
        wb = self._derived("bf16_cin8", lambda: ops.weight_bf16(wpad))
        want = bool(stats) and SCHED.fuse_bn_stats
        bn = stats if isinstance(stats, BatchNorm2d) else None
        res = ops.conv2d_fwd_bf16(x.t, wpad, wb, None, self.stride, self.padding, self.dilation, stats=want, bn=bn if want else None)
        if want:
            ya = Act(res[0])
            if bn is not None:
                ya.bn = res[1]
            else:
                ya.stats = res[1]
        else:
            ya = Act(res)
        if tape.enabled:
            def bwd():
                dy = ya.take_grad()
                if self.weight.requires_grad:
                    dwp = torch.empty_like(wpad)
                    ops.conv2d_wgrad_bf16(x.t, dy, wpad, dwp, self.stride, self.padding, self.dilation, beta=0)
                    g = _param_grad(self.weight)
                    dw = ops.nhwc_to_nchw(dwp.permute(0, 2, 3, 1), self.cin)          # [Cout,Cin,R,S]
                    ops.copy2d(dw, dw.numel(), g, g.numel(), 1, dw.numel(), beta=1)
            tape.push(bwd)
        return ya

    def _fwd_bf16(self, x, tape, out=None, stats=False):
        """bf16 activations in; bf16 out -- or fp32 out for the heads' final convs (`out_fp32`, or Cout % 8 != 0), whose
        results feed the fp32 loss tail.  Weights: fp32 master copy cast per launch; weight gradient fp32."""
        if self.act:
            raise RuntimeError("fused activation has no tape backward; use act=0 on trained paths")
        if self.cin % 8:
            return self._fwd_bf16_padded_cin(x, tape, out, stats)
        out_f32 = getattr(self, "out_fp32", False) or self.cout % 8 != 0
        if out_f32:
            w, b = self._packed_weight(), self._packed_bias()          # Cout padded to a multiple of 4: fp32 quads
            y = ops.conv2d_fwd_bf16(x.t, w, None, b, self.stride, self.padding, self.dilation, out=out,
                                    out_dtype=torch.float32)
            ya = Act(y)
        else:
            w, b = self.weight, self.bias
            if stats and SCHED.fuse_bn_stats and b is None and out is None:
                bn = stats if isinstance(stats, BatchNorm2d) else None
                y, sums = ops.conv2d_fwd_bf16(x.t, w, self._derived("bf16", lambda: ops.weight_bf16(w)), None, self.stride,
                                              self.padding, self.dilation, stats=True, bn=bn)
                ya = Act(y)
                if bn is not None:
                    ya.bn = sums
                else:
                    ya.stats = sums
            else:
                ya = Act(ops.conv2d_fwd_bf16(x.t, w, self._derived("bf16", lambda: ops.weight_bf16(w)), b, self.stride,
                                             self.padding, self.dilation, out=out))
        tok = object()
        if x.first_reader is None:
            x.first_reader = tok                   # first reader in forward order = last writer of x's gradient in the backward pass
        if tape.enabled:
            def bwd():
                dy = ya.take_grad()
                if self.bias is not None and self.bias.requires_grad:
                    if dy.shape[3] == self.cout:
                        ops.colsum(dy, _param_grad(self.bias), beta=1)
                    else:
                        db = torch.zeros(dy.shape[3], dtype=torch.float32, device=dy.device)
                        ops.colsum(dy, db, beta=0)
                        _param_grad(self.bias).add_(db[:self.cout])
                wk = self.weight
                if dy.dtype not in ops.H16:
                    # fp32 gradient of an fp32 output: as bf16 with the channels zero-padded to a multiple of 8
                    c8 = _round_up(dy.shape[3], 8)
                    d16 = (torch.zeros if c8 != dy.shape[3] else torch.empty)(tuple(dy.shape[:3]) + (c8,),
                                                                                dtype=x.t.dtype, device=dy.device)
                    ops.cast(dy, x.t.dtype, out=d16[..., :dy.shape[3]])
                    dy = d16
                    wk = self._packed_weight(8)
                done_dgrad = [False]

                def dgrad():
                    gx, beta = x.grad_slot()
                    wt = self._derived("bf16_t", lambda: ops.weight_bf16(wk, True)) if wk is self.weight else None
                    tile = 64 if self.cin <= 64 else (128 if self.cin <= 128 else 256)
                    if (SCHED.fold_masked_dgrad and x.relu_out and x.first_reader is tok and wt is not None and self.stride == 1
                            and self.cin % tile == 0 and gx.dtype in ops.H16 and x.t.dtype in ops.H16):
                        # x is the output of a folded conv + norm + ReLU and this is the LAST writer of its gradient: the mask of
                        # that ReLU and the column sums its backward needs come out of this epilogue (no pass over the gradient)
                        _, cs = ops.conv2d_dgrad_masked_bf16(dy, wk, x.t.shape, wt, x.t, gx, beta=beta, stride=self.stride,
                                                             pad=self.padding, dil=self.dilation, mask_bits=x.relu_bits)
                        x.grad_masked, x.grad_colsum = True, cs
                    else:
                        ops.conv2d_dgrad_bf16(dy, wk, x.t.shape, self.stride, self.padding, self.dilation, out=gx, beta=beta,
                                              w_t_bf16=wt)
                    done_dgrad[0] = True
                if self.weight.requires_grad:
                    padded = wk is not self.weight
                    dw = torch.empty_like(wk) if padded else _param_grad(self.weight)
                    if SCHED.overlap_wgrad and x.needs_grad and not padded:
                        side = _side_stream(dy.device)
                        if SCHED.late_wgrad:
                            dgrad()                   # see SCHED.late_wgrad: the weight gradient beside the next batch-norm backward
                        side.wait_stream(torch.cuda.current_stream(dy.device))
                        with torch.cuda.stream(side):
                            ops.conv2d_wgrad_bf16(x.t, dy, wk, dw, self.stride, self.padding, self.dilation, beta=1)
                        dy.record_stream(side)
                        x.t.record_stream(side)
                    else:
                        ops.conv2d_wgrad_bf16(x.t, dy, wk, dw, self.stride, self.padding, self.dilation,
                                              beta=0 if padded else 1)
                    if padded:                                           # prefix = the real rows
                        g = _param_grad(self.weight)
                        n = g.numel()
                        ops.copy2d(dw.permute(0, 2, 3, 1), n, g.permute(0, 2, 3, 1), n, 1, n, beta=1)
                if x.needs_grad and not done_dgrad[0]:
                    dgrad()
            tape.push(bwd)
        return ya

    def fwd(self, x, tape, out=None, stats=False):
        """Returns an Act with cout_pad channels (the padded ones are exactly zero).
        stats: the following layer is a training-mode batch norm; the conv epilogue produces its statistics.  Pass the
        BatchNorm2d itself (`bn.stats_request(dist)`) and the reduction of those statistics finalizes it in the same launch."""
        if x.t.dtype in ops.H16:
            return self._fwd_bf16(x, tape, out, stats)
        w, b = self._packed_weight(), self._packed_bias()
        ax = aw = wp = None
        # the input as fp16 pair planes (written by the batch norm that produced it): this conv then never splits an activation
        use_pl = x.planes is not None and out is None and self.planes_ok(tape.act_dtype) and ops.fused_stats_available()
        if x.split and not use_pl:
            raise ops._lib.RcfHipError("this activation exists as fp16 pair planes only and the conv cannot read them")
        xin = x.planes if use_pl else x.t
        tok = object()
        if x.first_reader is None:
            x.first_reader = tok                   # first reader in forward order = last writer of x's gradient in the backward pass
            # ... and as such it can take the identity branch's deferred gradient as a masked addend (checked again at launch)
            x.addend_ok = (SCHED.defer_residual and SCHED.fp16_pairs and tape.enabled and out is None and w is self.weight and
                           (self.cin in (64, 128) or self.cin % 256 == 0))
        yamax = None
        if SCHED.fp16_pairs:
            own = w is self.weight                                       # padded copies are rebuilt per call: not cached
            ax = x.range()
            aw = self._derived("amax", lambda: ops.absmax(ops.weight_rsck(w))) if own else ops.absmax(ops.weight_rsck(w))
            if use_pl or x.t.shape[0] * x.t.shape[1] * x.t.shape[2] >= 4096:      # many row tiles would each split the weights
                wp = self._derived("pairs", lambda: ops.weight_pairs(w, aw)) if own else ops.weight_pairs(w, aw)
            if stats and out is None and ops.fused_stats_available():
                yamax = ops.new_amax(xin.device)   # the output's range from the epilogue: the following batch norm's bound needs it
        if stats and SCHED.fuse_bn_stats and b is None and not self.act and out is None and self.cout_pad == self.cout \
                and ops.fused_stats_available():
            bn = stats if isinstance(stats, BatchNorm2d) else None
            y, sums = ops.conv2d_fwd_stats(xin, w, self.stride, self.padding, self.dilation, amax=(ax, aw), w_pairs=wp, bn=bn,
                                           x_planes=use_pl, amax_y=yamax)
            ya = Act(y)
            if bn is not None:
                ya.bn = sums
            else:
                ya.stats = sums
        else:
            y = ops.conv2d_fwd(xin, w, b, self.stride, self.padding, self.dilation, self.act, self.slope, out=out,
                               amax=(ax, aw), w_pairs=wp, x_planes=use_pl, amax_y=yamax)
            ya = Act(y)
        ya.amax = yamax
        ya.accepts_plane_grad = use_pl
        if tape.enabled:
            def bwd():
                ady = ya.take_grad_range() if SCHED.fp16_pairs else None    # shared by the data and the weight gradient
                dpl = ya.grad_is_planes                                # the batch norm's backward wrote dy as fp16 pair planes
                dy = ya.take_grad()
                if self.act:
                    raise RuntimeError("fused activation has no tape backward; use act=0 on trained paths")
                if use_pl and x.split and not dpl:
                    raise ops._lib.RcfHipError("weight gradient of a conv whose input exists as pair planes only needs its output "
                                               "gradient in the same format (the batch norm behind it did not write it so)")
                done_dgrad = [False]

                def late_dgrad():
                    wpt = None
                    if SCHED.fp16_pairs and w is self.weight and ady is not None and aw is not None:
                        wpt = self._derived("pairs_t", lambda: ops.weight_pairs_t(w, aw))
                    can_add = (x.pending_add is not None and x.first_reader is tok and wpt is not None and
                               ops.dgrad_takes_addend(w, x.t.shape, self.stride, self.padding, self.dilation, ops.pitch_of(dy),
                                                      (ady, aw), wpt, dpl))
                    gx, beta = x.grad_slot(takes_addend=can_add)
                    add = x.take_pending() if (can_add and beta == 0) else None
                    # the range of dx comes out of the epilogue (after the accumulation when beta = 1): exact for the whole tensor
                    gamax = ops.new_amax(dy.device) if SCHED.fp16_pairs and wpt is not None and ops.fused_stats_available() else None
                    bnb = x.bn_ctx if (SCHED.fuse_bn_bwd and wpt is not None and x.first_reader is tok) else None
                    r = ops.conv2d_dgrad(dy, w, x.t.shape, self.stride, self.padding, self.dilation, out=gx, beta=beta,
                                         amax=(ady, aw), w_pairs_t=wpt, dy_planes=dpl, amax_y=gamax, bn_bwd=bnb, addend=add)
                    x.grad_amax = gamax
                    if bnb is not None:
                        x.grad_sums2 = r[1]            # None when the launch had no such epilogue: the norm runs its reduction pass
                    done_dgrad[0] = True
                xw, xpl = (x.planes, True) if (dpl and use_pl) else (x.t, False)      # the weight gradient's x in dy's format
                if self.weight.requires_grad:
                    if self.cin_pad == self.cin and self.cout_pad == self.cout:
                        if SCHED.overlap_wgrad and x.needs_grad:
                            side = _side_stream(dy.device)
                            if SCHED.late_wgrad:
                                # the data gradient first, alone; the weight gradient starts when it is done and so runs beside
                                # the NEXT layer's batch-norm backward (HBM-bound) instead of beside this layer's data
                                # gradient (MFMA-bound like itself)
                                late_dgrad()
                            side.wait_stream(torch.cuda.current_stream(dy.device))       # dy is ready (SCHED.late_wgrad: the data gradient is done)
                            with torch.cuda.stream(side):
                                ops.conv2d_wgrad(xw, dy, w, _param_grad(self.weight), self.stride, self.padding,
                                                 self.dilation, beta=1, amax=(ax, ady), small_tile=SCHED.late_wgrad, planes=xpl)
                            dy.record_stream(side)
                            xw.record_stream(side)
                        else:
                            ops.conv2d_wgrad(xw, dy, w, _param_grad(self.weight), self.stride, self.padding,
                                             self.dilation, beta=1, amax=(ax, ady), planes=xpl)
                    else:
                        dwp = torch.empty_like(w)
                        ops.conv2d_wgrad(x.t, dy, w, dwp, self.stride, self.padding, self.dilation, beta=0,
                                         amax=(ax, ady))
                        g = _param_grad(self.weight)
                        if self.cin_pad != self.cin:
                            dw = ops.nhwc_to_nchw(dwp.permute(0, 2, 3, 1), self.cin)      # [Cout,Cin,R,S]
                            ops.copy2d(dw, dw.numel(), g, g.numel(), 1, dw.numel(), beta=1)
                        else:
                            n = g.numel()                                                   # prefix = real rows
                            ops.copy2d(dwp.permute(0, 2, 3, 1), n, g.permute(0, 2, 3, 1), n, 1, n, beta=1)
                if self.bias is not None and self.bias.requires_grad:
                    if self.cout_pad == self.cout:
                        ops.colsum(dy, _param_grad(self.bias), beta=1)
                    else:
                        db = torch.zeros(self.cout_pad, dtype=torch.float32, device=dy.device)
                        ops.colsum(dy, db, beta=0)
                        _param_grad(self.bias).add_(db[:self.cout])
                if x.needs_grad and not done_dgrad[0]:
                    late_dgrad()
            tape.push(bwd)
        return ya


def _residual_norm_sums(residual, dy, y, relu, rmask, chan_scale):
    """local backward sums [sum g | sum g xhat] of the plain norm that produced `residual` (Act.bn_in), g = the join's output
    gradient dy under the join's own ReLU mask -- what that norm's backward would reduce from its output gradient; None when the
    residual is not such a norm's output or somebody else also feeds its gradient"""
    if (residual is None or not residual.needs_grad or residual.bn_in is None or residual.grad is not None or
            residual.pending_add is not None or chan_scale is not None or not relu or (rmask is None and y is None)):
        return None
    xr, mean_r, invstd_r = residual.bn_in
    if xr.dtype != dy.dtype or tuple(xr.shape) != tuple(dy.shape):
        return None
    return ops.bn_bwd_reduce(dy, xr, y, mean_r, invstd_r, True, relu_mask=rmask)


class BatchNorm2d(nn.Module):
    """(Sync)BatchNorm2d, eps 1e-5, momentum 0.1, with fused ReLU / residual add / Dropout2d scale.
    Same parameter and buffer names as torch's (state-dict compatible)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, requires_grad=True, sync=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.sync = sync                 # SyncBN: statistics over the global batch (every rank holds the same number of rows)
        self.weight = nn.Parameter(torch.ones(num_features), requires_grad=requires_grad)
        self.bias = nn.Parameter(torch.zeros(num_features), requires_grad=requires_grad)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def stats_request(self, dist=None):
        """what to hand the producing conv as `stats`: False in eval mode; True when the statistics must be all-reduced
        first (SyncBN in a distributed run); otherwise this module -- the conv's reduction finalizes it"""
        if not self.training:
            return False
        if self.sync and dist is not None and dist.on:
            return True
        return self if SCHED.fuse_bn_finalize else True

    def fwd(self, x, tape, relu, residual=None, chan_scale=None, out=None, dist=None, planes=None, lazy=False):
        """planes: "only" / "both" -- the output is (also) wanted as fp16 pair planes for the conv(s) that read it (Act.planes;
        "only": no fp32 copy is written); honoured when the bound of the output is computable (fp32 step, the range of x known
        from the producing conv's epilogue, no dropout scale) -- otherwise the fp32 output alone, as without the request.
        lazy (a plain norm whose only reader is a join taking it as the residual -- a stage's downsample branch): statistics,
        finalize and backward as always, but NO apply pass: the Act returned holds the norm's INPUT and `res_norm`, and the join's
        own pass normalises it on the way in (bit-identical: the same operations, rounded to the storage type in between)."""
        xt = x.t
        lazy = lazy and SCHED.lazy_downsample_norm and not relu and residual is None and chan_scale is None and out is None and planes is None
        if not self.sync:
            dist = None
        if self.training and x.bn is not None:            # finalized by the producing conv's statistics reduction
            mean, invstd, count = x.bn
        elif self.training:
            local_rows = xt.shape[0] * xt.shape[1] * xt.shape[2]
            sums = x.stats if x.stats is not None else ops.bn_stats(xt)
            count = local_rows
            if dist is not None and dist.on:                    # SyncBN: statistics over the global batch
                if not x.stats_global:
                    dist.allreduce_sum(sums)
                count = local_rows * dist.world
            mean, invstd = ops.bn_finalize(sums, count, self.eps, self.momentum, self.running_mean, self.running_var)
            self.num_batches_tracked += 1
        else:
            count = 0
            mean, invstd = self.running_mean, ops.bn_invstd_from_var(self.running_var, self.eps)
        rmask = None
        if relu and tape.enabled and SCHED.relu_bitmask:
            # the backward kernels read this (one byte per 4 channels) instead of y
            rmask = torch.empty(xt.numel() // 4, dtype=torch.uint8, device=xt.device)
        ydt = out.dtype if out is not None else (tape.act_dtype if xt.dtype == torch.float32 else xt.dtype)
        yamax = ops.new_amax(xt.device) if SCHED.fp16_pairs and ydt == torch.float32 else None
        pl = (planes if SCHED.planes and SCHED.fp16_pairs and planes in ("only", "both") and out is None and chan_scale is None and
              xt.dtype == torch.float32 and ydt == torch.float32 and x.amax is not None and xt.shape[3] % 8 == 0 and
              xt.is_contiguous() and (residual is None or residual.t.dtype == torch.float32) else None)
        if pl == "only" and (RELU_TRACE is not None or (relu and tape.enabled and rmask is None)):
            pl = "both"                               # somebody reads the fp32 output itself
        pbuf = torch.empty(tuple(xt.shape), dtype=torch.float32, device=xt.device) if pl else None
        if lazy and ydt == xt.dtype:
            y = None
            ya = Act(xt)
            ya.res_norm = (mean, invstd, self.weight.detach(), self.bias.detach())
            ya.res_ctx = (self, x, count)
            ya.amax = x.amax                          # the range of the RAW tensor (what rcf_bn_apply_res_mp's bound starts from)
        else:
            y = ops.bn_apply(xt, mean, invstd, self.weight, self.bias, relu,
                             residual=residual.t if residual is not None else None, chan_scale=chan_scale, out=out,
                             relu_mask=rmask, amax_out=yamax, out_dtype=ydt, planes=pbuf, planes_only=pl == "only",
                             amax_x=x.amax if pl else None, amax_res=residual.range() if (pl and residual is not None) else None,
                             res_norm=residual.res_norm if residual is not None else None)
            ya = Act(y if y is not None else pbuf)
            ya.amax = yamax                           # with planes: the bound they are scaled by (a valid range for any reader)
            ya.planes, ya.split = pbuf, pl == "only"
        if RELU_TRACE is not None and relu:
            RELU_TRACE.append(y > 0)
        if tape.enabled:
            if not self.training:
                raise RuntimeError("tape backward through eval-mode BN is not implemented")
            if not relu and residual is None and chan_scale is None and dist is not None and dist.on:
                ya.bn_in = (xt, mean, invstd)
            if relu and rmask is not None and chan_scale is None and xt.dtype == torch.float32 and ydt == torch.float32 and xt.is_contiguous():
                ya.bn_ctx = (xt, rmask, mean, invstd)

            planes_at_fwd = bool(SCHED.planes)

            def bwd():
                if ya.res_done:                       # the join ran this (lazy downsample) norm's backward beside its own
                    ya.res_done = False
                    return
                # dx as fp16 pair planes when the conv that produced x takes its output gradient so (its data and weight gradient
                # are the only readers); the bound needs the ranges of x and of dy
                # (decided with the FORWARD pass's setting: a conv whose input exists as planes only cannot take an fp32 gradient,
                # whatever the schedule object says by the time the backward pass runs)
                relu_b, rmask_b, y_b = relu, rmask, y
                pend = None
                if ya.res_norm is not None and ya.pending_add is not None and ya.grad is None:
                    # lazy downsample norm: the join left (its output gradient, its sign bits) instead of writing this norm's
                    # output gradient -- the two passes below read the one under the other, as they would for a norm + ReLU
                    pend = ya.take_pending()
                if pend is not None:
                    dy, rmask_b = pend
                    relu_b, y_b = True, None
                    dpl = (x.accepts_plane_grad and planes_at_fwd and xt.dtype == torch.float32 and x.amax is not None
                           and dy.dtype == torch.float32)
                    ady, ya.grad_amax = ya.grad_amax, None       # the join's: the mask only lowers the range
                    if dpl and ady is None:
                        ady = ops.absmax(dy)
                else:
                    dpl = (x.accepts_plane_grad and planes_at_fwd and xt.dtype == torch.float32 and x.amax is not None
                           and (not relu or rmask is not None) and ya.grad is not None and ya.grad.dtype == torch.float32)
                    ady = ya.take_grad_range() if dpl else None
                    dy = ya.take_grad()
                s2, ya.grad_sums2 = ya.grad_sums2, None          # from the epilogue of the data gradient that wrote dy last
                s2_local = None
                if (s2 is None and pend is None and SCHED.merge_downsample_bwd and residual is not None and residual.res_ctx is not None
                        and residual.needs_grad and relu and rmask is not None and chan_scale is None and residual.grad is None
                        and residual.pending_add is None and dy.dtype == xt.dtype == residual.t.dtype and dy.is_contiguous()):
                    # a stage's first block: this join and its lazy downsample norm take their backward of the same masked gradient --
                    # ONE reduction and ONE apply pass over dy and the sign bits for both (rcf_bn_bwd_reduce2_mp / _apply2_mp)
                    bn_d, x_d, count_d = residual.res_ctx
                    mean_d, invstd_d = residual.res_norm[0], residual.res_norm[1]
                    dpl_d = (x_d.accepts_plane_grad and planes_at_fwd and x_d.t.dtype == torch.float32 and x_d.amax is not None
                             and dy.dtype == torch.float32)
                    if dpl_d == dpl and count_d == count and x_d.t.is_contiguous() and xt.is_contiguous():
                        C = xt.shape[3]
                        sums4 = ops.bn_bwd_reduce2(dy, xt, x_d.t, mean, invstd, mean_d, invstd_d, rmask)
                        local4 = None
                        if dist is not None and dist.on:
                            local4 = sums4.clone()       # dgamma / dbeta stay per rank; the gradient all-reduce adds them
                            dist.allreduce_sum(sums4)    # one exchange for both norms
                        gx, _ = x.grad_slot()
                        gxd, _ = x_d.grad_slot()
                        fp = SCHED.fp16_pairs and xt.dtype == torch.float32
                        gamax = x.grad_amax = ops.new_amax(dy.device) if fp else None
                        gamax_d = x_d.grad_amax = ops.new_amax(dy.device) if fp else None
                        ops.bn_bwd_apply2(dy, xt, mean, invstd, self.weight, rmask, sums4[:2 * C], count,
                                          _param_grad(self.weight) if self.weight.requires_grad else None,
                                          _param_grad(self.bias) if self.bias.requires_grad else None, gx,
                                          x_d.t, mean_d, invstd_d, bn_d.weight, sums4[2 * C:],
                                          _param_grad(bn_d.weight) if bn_d.weight.requires_grad else None,
                                          _param_grad(bn_d.bias) if bn_d.bias.requires_grad else None, gxd,
                                          sums2_local=local4[:2 * C] if local4 is not None else None,
                                          sums2_2_local=local4[2 * C:] if local4 is not None else None,
                                          amax_out=gamax, amax_out2=gamax_d, dx_planes=dpl, amax_x=x.amax if dpl else None,
                                          amax_x2=x_d.amax if dpl else None, amax_dy=ady)
                        x.grad_is_planes = x_d.grad_is_planes = dpl
                        residual.res_done = True
                        return
                if isinstance(s2, tuple):                        # (global, local): exchanged already, together with the join's own
                    s2, s2_local = s2
                else:
                    if s2 is None:
                        s2 = ops.bn_bwd_reduce(dy, xt, y_b, mean, invstd, relu_b, chan_scale=chan_scale, relu_mask=rmask_b)
                    if dist is not None and dist.on:
                        s2_local = s2.clone()            # dgamma/dbeta stay per-rank; the gradient all-reduce adds them
                        r_s2 = _residual_norm_sums(residual, dy, y_b, relu_b, rmask_b, chan_scale)
                        if r_s2 is not None:
                            # a stage's first block: the downsample norm's backward sums depend on this join's masked output
                            # gradient only -- one exchange for both norms instead of two latency-bound ones
                            r_local = r_s2.clone()
                            dist.allreduce_sum_many([s2, r_s2])
                            residual.grad_sums2 = (r_s2, r_local)
                        else:
                            dist.allreduce_sum(s2)
                dres, rbeta = (None, 0)
                if residual is not None and residual.needs_grad:
                    if (residual.res_norm is not None and relu and rmask is not None and chan_scale is None and residual.grad is None
                            and residual.pending_add is None and dy.dtype == residual.t.dtype and dy.is_contiguous()):
                        residual.pending_add = (dy, rmask)       # the lazy downsample norm's backward reads dy under this mask itself
                        residual.grad_amax = ady
                    elif (SCHED.defer_residual and relu and rmask is not None and chan_scale is None and residual.addend_ok and residual.grad is None
                            and residual.pending_add is None and dy.dtype == torch.float32 and residual.t.dtype == torch.float32
                            and dy.is_contiguous()):
                        residual.pending_add = (dy, rmask)       # the identity's gradient = dy under this join's mask: not written here
                    else:
                        dres, rbeta = residual.grad_slot()
                        if rbeta == 0 and ady is not None and chan_scale is None:
                            # first writer: what it writes is dy under the ReLU mask, so dy's range bounds it (saves the
                            # downsample norms' backward a pass over their 4C-wide output gradient)
                            residual.grad_amax = ady
                gx, _ = x.grad_slot()     # conv outputs feed exactly one BN: always first writer
                gamax = x.grad_amax = ops.new_amax(dy.device) if SCHED.fp16_pairs and xt.dtype == torch.float32 else None
                ops.bn_bwd_apply(dy, xt, y_b, mean, invstd, self.weight, relu_b, s2, count,
                                 _param_grad(self.weight) if self.weight.requires_grad else None,
                                 _param_grad(self.bias) if self.bias.requires_grad else None,
                                 dx=gx, dres=dres, res_beta=rbeta, chan_scale=chan_scale, sums2_local=s2_local,
                                 relu_mask=rmask_b, amax_out=gamax, dx_planes=dpl, amax_x=x.amax if dpl else None, amax_dy=ady)
                x.grad_is_planes = dpl
            tape.push(bwd)
        return ya


def fold_ok(conv, bn, x, residual=None):
    """can conv (1x1, stride 1) -> bn (training mode) [-> + residual] [-> ReLU] run as conv_bn_fold?"""
    t = x.t
    return (SCHED.fold_bn and t.dtype in ops.H16 and conv.k == 1 and conv.stride == 1 and conv.padding == 0 and conv.bias is None
            and not conv.act and bn.training and conv.cin % 64 == 0 and conv.cin <= SCHED.fold_max_k
            and (conv.cout in (64, 128) or conv.cout % 256 == 0) and ops.relu_mask_colsum_ok(conv.cout)
            and t.is_contiguous() and bn.num_features == conv.cout
            and (residual is None or (residual.t.dtype == t.dtype and tuple(residual.t.shape[:3]) == tuple(t.shape[:3])
                                      and residual.t.shape[3] == conv.cout)))


def conv_bn_fold(conv, bn, x, tape, relu, residual=None, dist=None):
    """y = [relu](bn(conv(x)) [+ residual]) for a 1x1 stride-1 conv and a training-mode batch norm, in ONE pass over x: the norm's
    statistics come from the moments of x (column sums + Gram matrix), the conv kernel normalises, adds and clamps in its
    epilogue; the backward pass needs neither the conv's output nor its gradient (csrc/foldbn.hip has the algebra).
    models/resnet.py:281-296, models/res_layer.py:53-63."""
    xt, w = x.t, conv.weight
    rows, K, C = xt.shape[0] * xt.shape[1] * xt.shape[2], conv.cin, conv.cout
    if not bn.sync:
        dist = None
    wb = conv._derived("bf16", lambda: ops.weight_bf16(w))
    S = ops.gram_bf16(xt)
    A1 = ops.bn_stats(xt)                                   # fp64 [2K]: the column sums (and sums of squares, unused) of x
    count = rows
    if dist is not None and dist.on:                        # SyncBN: the statistics of the global batch
        P, sums = ops.fold_fwd(S, A1, w, rows=rows)
        dist.allreduce_sum(sums)
        count = rows * dist.world
        mean, invstd, scale, shift = ops.fold_finalize(sums, count, bn)
    else:
        P, (mean, invstd, scale, shift) = ops.fold_fwd(S, A1, w, bn, count)
    if x.first_reader is None:
        x.first_reader = object()
    want_bits = bool(relu) and tape.enabled and SCHED.fold_masked_dgrad
    y = ops.conv2d_fwd_affine_bf16(xt, w, wb, scale, shift, residual.t if residual is not None else None, relu, want_bits=want_bits)
    bits = None
    if want_bits:
        y, bits = y
    ya = Act(y)
    ya.relu_out = bool(relu)
    ya.relu_bits = bits
    if RELU_TRACE is not None and relu:
        RELU_TRACE.append(y > 0)
    if tape.enabled:
        def bwd():
            masked, cs = ya.grad_masked, ya.grad_colsum
            dy = ya.take_grad()
            if relu and not masked:
                g, cs = ops.relu_mask_colsum(dy, y, out=dy)       # in place: dy has no other reader
            else:
                g = dy
                if cs is None:
                    cs = ops.bn_stats(g)                         # no ReLU: the plain column sums
            if residual is not None and residual.needs_grad:
                if residual.grad is None and residual.pending_add is None:
                    residual.grad = g        # the identity's gradient IS g; whoever writes next accumulates, after the reads below
                    residual.grad_colsum = cs    # ... and these are its column sums until then (a folded downsample norm wants them)
                else:
                    rg, _ = residual.grad_slot()
                    ops.copy2d(g, ops.pitch_of(g), rg, ops.pitch_of(rg), rows, C, beta=1)
            # dx = g (a W)^T - x T + c0; T and c0 need G = g^T x.  Both products stay on THIS stream: G is on the critical path, and
            # waiting for the second stream here would also wait for every weight gradient queued on it (measured: +0.7 ms per step)
            G = torch.empty((C, K, 1, 1), dtype=torch.float32, device=xt.device)
            ops.conv2d_wgrad_bf16(xt, g, w, G, 1, 0, 1, beta=0)
            gx = None
            if x.needs_grad:
                gx, beta = x.grad_slot()
                wg_t = ops.fold_wg(w, scale)
                ops.conv2d_dgrad_bf16(g, w, xt.shape, 1, 0, 1, out=gx, beta=beta, w_t_bf16=wg_t)      # g (a W)^T
            sums2 = ops.fold_bwd_sums(G, w, cs, mean, invstd)
            s2_local = None
            if dist is not None and dist.on:
                s2_local = sums2.clone()             # dgamma / dbeta stay per rank; the gradient all-reduce adds them
                r_s2 = None
                if residual is not None and residual.needs_grad and residual.bn_in is not None and residual.grad is g:
                    # a stage's first block: the (plain) downsample norm's backward sums from the same g, in the same exchange
                    xr, mean_r, invstd_r = residual.bn_in
                    if xr.dtype == g.dtype:
                        r_s2 = ops.bn_bwd_reduce(g, xr, None, mean_r, invstd_r, False)
                if r_s2 is not None:
                    r_local = r_s2.clone()
                    dist.allreduce_sum_many([sums2, r_s2])
                    residual.grad_sums2 = (r_s2, r_local)
                else:
                    dist.allreduce_sum(sums2)
            negT, c0 = ops.fold_bwd_prepare(
                G, P, A1, w, sums2, s2_local, count, mean, invstd, bn.weight, _param_grad(w) if w.requires_grad else None,
                _param_grad(bn.weight) if bn.weight.requires_grad else None, _param_grad(bn.bias) if bn.bias.requires_grad else None)
            if gx is not None:
                ops.conv2d_fwd_bf16(xt, S, negT, c0, 1, 0, 1, out=gx, beta=1)                      # - x T + c0
        tape.push(bwd)
    return ya


def maxpool3x3s2(x, tape):
    y, am = ops.maxpool_fwd(x.t)
    ya = Act(y)
    ya.amax = x.amax            # every output is one of the inputs: the input's range bounds the output (no pass over it)

    def bwd():
        dy = ya.take_grad()
        if x.needs_grad:
            assert x.grad is None
            x.grad = ops.maxpool_bwd(dy, am, x.t.shape)
    tape.push(bwd)
    return ya


def resize_into(x, tape, size, align_corners, out):
    """Bilinear resize of x written into `out` (possibly a channel slice); returns nothing: the caller
    owns the Act of the enclosing buffer and routes the slice gradient back through `bwd_from`."""
    ops.resize_nhwc_fwd(x.t, size, align_corners, out=out)


def resize_act(x, tape, size, align_corners=False):
    """bilinear resize of an activation with its backward (mmseg.ops.resize on a tensor that needs a gradient,
    models/rcf_model.py:421-422)"""
    y = Act(ops.resize_nhwc_fwd(x.t, tuple(size), align_corners))

    def bwd():
        g = y.take_grad()
        if x.needs_grad:
            gx, beta = x.grad_slot()
            ops.resize_nhwc_bwd(g, x.t.shape[1:3], align_corners, out=gx, beta=beta)
    tape.push(bwd)
    return y


def concat_channels(parts, tape, size=None, align_corners=False, widths=None):
    """resize_concat of models/decode_head.py:151-164: every part is bilinearly resized to `size`
    (default: the first part's) and written into its channel slice of one NHWC buffer.
    `widths`: take only the first widths[i] channels of part i (drops zero-padded output channels)."""
    N, H, W, _ = parts[0].t.shape
    if size is not None:
        H, W = size
    widths = [p.t.shape[3] for p in parts] if widths is None else list(widths)
    ctot = sum(widths)
    buf = torch.empty((N, H, W, ctot), dtype=parts[0].t.dtype, device=parts[0].t.device)
    offs, o = [], 0
    for p, c in zip(parts, widths):
        sl = buf[..., o:o + c]
        if tuple(p.t.shape[1:3]) == (H, W):
            ops.copy2d(p.t, ops.pitch_of(p.t), sl, ctot, N * H * W, c)
        else:
            assert c == p.t.shape[3]
            ops.resize_nhwc_fwd(p.t, (H, W), align_corners, out=sl)
        offs.append(o)
        o += c
    ya = Act(buf)

    def bwd():
        g = ya.take_grad()
        for p, o, c in zip(parts, offs, widths):
            if not p.needs_grad:
                continue
            gs = g[..., o:o + c]
            gp, beta = p.grad_slot()
            if c < p.t.shape[3] and beta == 0:
                ops.fill(gp, 0.0)                  # the dropped (padded) channels carry no gradient
            if tuple(p.t.shape[1:3]) == (H, W):
                ops.copy2d(gs, ctot, gp, ops.pitch_of(gp), N * H * W, c, beta=beta)
            else:
                ops.resize_nhwc_bwd(gs, p.t.shape[1:3], align_corners, out=gp, beta=beta)
    tape.push(bwd)
    return ya


def commuted_concat_conv_ok(a, b, conv, align_corners):
    """conv3x3(dilation d even, pad d, stride 1) over concat(a, bilinear_2x(b)): can the 2x upsampling be commuted?"""
    h, w = a.t.shape[1:3]
    d = conv.dilation
    return (conv.k == 3 and conv.stride == 1 and d % 2 == 0 and d >= 2 and conv.padding == d and conv.bias is None
            and not align_corners and (h, w) == (2 * b.t.shape[1], 2 * b.t.shape[2])
            and conv.cin == a.t.shape[3] + b.t.shape[3] and conv.cout_pad == conv.cout and conv.cin_pad == conv.cin
            and min(h, w) > 2 * (2 * d + 1) and a.t.is_contiguous() and b.t.is_contiguous() and ops.fused_stats_available()
            and (a.t.dtype == torch.float32 or (a.t.shape[3] % 8 == 0 and b.t.shape[3] % 8 == 0 and conv.cout % 8 == 0)))


def commuted_concat_conv(a, b, conv, tape):
    """y = conv(concat(a, U)), U = bilinear_2x(b) (decode_head2's first conv: models/decode_head.py:151-164 +
    models/fcn_head.py:107-118) without running the wide conv over the up-sampled channels.

    Bilinear 2x up-sampling (align_corners=False) is the same 2x2 stencil at every pixel of a given parity, and a
    dilation-d (d even) tap moves by d/2 source pixels without changing parity, so
        conv_d(U; Wb) == bilinear_2x(conv_{d/2}(b; Wb))
    wherever neither the zero padding of the conv nor the edge clamp of the resize is involved: everywhere except a
    band of d+1 pixels along the border (checked numerically in tests).  So: the up-sampled channels are convolved at
    HALF resolution (1/4 of the FLOPs) and up-sampled afterwards, the border band is computed directly from the
    up-sampled channels with the region-restricted kernels (the up-sampling itself is only materialised on the frame
    the band's taps reach), and `a`'s channels are an ordinary 256-channel conv added on top.
    The backward is the adjoint of exactly these pieces.  Same arithmetic up to fp32 summation order."""
    N, h, w, Ca = a.t.shape
    hb, wb, Cb = b.t.shape[1], b.t.shape[2], b.t.shape[3]
    d, C, Co = conv.dilation, Ca + Cb, conv.cout
    bw = d + 1                                   # border band (output pixels computed directly)
    bi = bw + d                                  # input pixels the band's gradient reaches
    W = conv.weight
    dt = a.t.dtype
    bf = dt in ops.H16                           # 16-bit-operand kernels (mixed-precision step), else fp32-level kernels
    assert b.t.dtype == dt
    wa = W[:, :Ca].contiguous(memory_format=torch.channels_last)
    wbt = W[:, Ca:].contiguous(memory_format=torch.channels_last)
    # the up-sampled channels are only ever needed on the frame the band's taps reach (bi = bw + d pixels)
    Uup = torch.empty((N, h, w, Cb), dtype=dt, device=a.t.device)
    ops.resize_nhwc_fwd(b.t, (h, w), False, out=Uup, frame=bi)
    interior = (bw, bw, h - 2 * bw, w - 2 * bw)
    band = (0, 0, h, w, bw)                      # the border frame of thickness bw, one launch
    # operand ranges of the fp16-pair kernels: bilinear interpolation is a convex combination (|Uup| <= max |b|),
    # and one range serves both halves of the weight
    ra = rb = rw = pa = pb = None
    h2 = SCHED.fp16_pairs and not bf
    if h2:
        ra, rb, rw = a.range(), b.range(), ops.absmax(ops.weight_rsck(W))
        pa, pb = ops.weight_pairs(wa, rw), ops.weight_pairs(wbt, rw)
    if bf:
        pa, pb = ops.weight_bf16(wa), ops.weight_bf16(wbt)

    def cfwd(x, wt, wpre, dil, rx, out=None, beta=0, region=None):
        if bf:
            return ops.conv2d_fwd_bf16(x, wt, wpre, None, 1, dil, dil, out=out, beta=beta, region=region)
        return ops.conv2d_fwd(x, wt, None, 1, dil, dil, out=out, beta=beta, region=region, amax=(rx, rw), w_pairs=wpre)

    def cwgrad(x, dy, wt, dw, dil, beta, rx, rdy, region=None):
        if bf:
            return ops.conv2d_wgrad_bf16(x, dy, wt, dw, 1, dil, dil, beta=beta, region=region)
        return ops.conv2d_wgrad(x, dy, wt, dw, 1, dil, dil, beta=beta, region=region, amax=(rx, rdy))

    wpt = {}                                     # transposed fp16-pair planes of the two weight halves, made once per backward

    def cdgrad(dy, wt, xshape, dil, out, beta, rdy, region=None):
        if bf:
            return ops.conv2d_dgrad_bf16(dy, wt, xshape, 1, dil, dil, out=out, beta=beta, region=region)
        pt = None
        if h2 and rdy is not None:
            pt = wpt.get(id(wt))
            if pt is None:
                pt = wpt[id(wt)] = ops.weight_pairs_t(wt, rw)
        return ops.conv2d_dgrad(dy, wt, xshape, 1, dil, dil, out=out, beta=beta, region=region, amax=(rdy, rw), w_pairs_t=pt)

    Z = cfwd(b.t, wbt, pb, d // 2, rb)
    y = ops.resize_nhwc_fwd(Z, (h, w), False)                              # interior: conv_d(up2(b)) = up2(conv_{d/2}(b))
    cfwd(Uup, wbt, pb, d, rb, out=y, beta=0, region=band)                  # band: directly
    cfwd(a.t, wa, pa, d, ra, out=y, beta=1)                                # a's channels: everywhere
    ya = Act(y)

    def bwd():
        rdy = ya.take_grad_range() if h2 else None                         # from the batch norm's backward when it left one
        dy = ya.take_grad()
        dy_int, dy_band = ops.split_rect(dy, interior)
        g4 = ops.resize_nhwc_bwd(dy_int, (hb, wb), False)                  # adjoint of the interior's up-sampling
        rg4 = None
        if h2:
            rg4 = ops.absmax(g4)                                           # dy_band is a part of dy
        if W.requires_grad:
            gW = _param_grad(W).permute(0, 2, 3, 1)                       # [Co,3,3,C] as stored
            dwa = torch.empty_like(wa)
            cwgrad(a.t, dy, wa, dwa, d, 0, ra, rdy)
            ops.copy2d(dwa.permute(0, 2, 3, 1), Ca, gW[..., :Ca], C, Co * 9, Ca, beta=1)
            dwb = torch.empty_like(wbt)
            cwgrad(b.t, g4, wbt, dwb, d // 2, 0, rb, rg4)
            cwgrad(Uup, dy, wbt, dwb, d, 1, rb, rdy, region=band)
            ops.copy2d(dwb.permute(0, 2, 3, 1), Cb, gW[..., Ca:], C, Co * 9, Cb, beta=1)
        if a.needs_grad:
            ga, beta = a.grad_slot()
            cdgrad(dy, wa, a.t.shape, d, ga, beta, rdy)
        if b.needs_grad:
            gb, beta = b.grad_slot()
            dUup = torch.empty_like(Uup)                                   # written (and read) on the bi-frame only
            cdgrad(dy_band, wbt, Uup.shape, d, dUup, 0, rdy, region=(0, 0, h, w, bi))
            # the whole-tensor term first, so that the frame term always accumulates (its kernel then only walks the
            # input pixels that can see the frame)
            cdgrad(g4, wbt, b.t.shape, d // 2, gb, beta, rg4)
            ops.resize_nhwc_bwd(dUup, (hb, wb), False, out=gb, beta=1, frame=bi)
    tape.push(bwd)
    return ya


def pair_concat(x, tape, B, I, order=None):
    """[B*I,h,w,C] -> [B,h,w,I*C]: frames of a pair side by side on channels
    (unflatten(0,(B,I)).flatten(1,2) of models/rcf_model.py:325 in NHWC).  `order` permutes the frames
    first (the [1, 0] swap of pred_joint_residual, models/rcf_model.py:342)."""
    N, H, W, C = x.t.shape
    assert N == B * I and x.t.is_contiguous()
    order = list(range(I)) if order is None else list(order)
    out = torch.empty((B, H, W, I * C), dtype=x.t.dtype, device=x.t.device)
    HW = H * W
    # one launch for the B x I copies when the frame order is an arithmetic progression (identity, or the swap of a pair)
    step = order[1] - order[0] if I > 1 else 1
    batched = all(order[i] == order[0] + i * step for i in range(I))
    if batched:
        ops.copy2d_batched(x.t[order[0]], C, (I * HW * C, step * HW * C), out, I * C, (HW * I * C, C), HW, C, (B, I))
    else:
        for b in range(B):
            for i in range(I):
                ops.copy2d(x.t[b * I + order[i]], C, out[b][..., i * C:], I * C, HW, C)
    ya = Act(out)
    ya.amax = x.amax                 # the same values in another order: the range the producer left is this tensor's range
    if x.planes is not None and batched and C % 2 == 0 and SCHED.planes:
        # the fp16 pair planes travel too ([pixel][h: C fp16 | m: C fp16] = C floats per pixel, h first): the pair's h halves side
        # by side, then its m halves -- two launches of C / 2 floats per pixel each, same bound
        pl = torch.empty((B, H, W, I * C), dtype=torch.float32, device=x.t.device)
        src = x.planes[order[0]]
        for half in (0, 1):
            ops.copy2d_batched(src.reshape(-1)[half * (C // 2):], C, (I * HW * C, step * HW * C),
                               pl.reshape(-1)[half * (I * C // 2):], I * C, (HW * I * C, C // 2), HW, C // 2, (B, I))
        ya.planes = pl

    def bwd():
        g = ya.take_grad()
        if not x.needs_grad:
            return
        gx, beta = x.grad_slot()
        if batched:
            ops.copy2d_batched(g, I * C, (HW * I * C, C), gx[order[0]], C, (I * HW * C, step * HW * C), HW, C, (B, I),
                               beta=beta)
        else:
            for b in range(B):
                for i in range(I):
                    ops.copy2d(g[b][..., i * C:], I * C, gx[b * I + order[i]], C, HW, C, beta=beta)
    tape.push(bwd)
    return ya
