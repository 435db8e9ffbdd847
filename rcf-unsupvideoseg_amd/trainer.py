"""Minimal trainer for the hot path: flat parameter/gradient buffers, fused Adam, poly LR,
data-parallel gradient all-reduce.  Stands in for main.py:158-178 (training_step), :294-310
(get_lr / configure_optimizers) and Lightning's DDP wrapper (main.py:453-455).

One process per GPU; `torch.distributed` (backend "nccl" == RCCL over xGMI) carries exactly two
kinds of traffic: the SyncBN statistics (inside the model) and the all-reduce of the flat fp32 gradient
buffer (157.9 MB for the stage-1 model), issued asynchronously in 6 contiguous chunks as backward finishes
them (heads, layer4 ... layer1, stem: the tape fires a mark after each group) so that all but the last
small chunk overlap with the rest of backward -- instead of DDP's 25 MB buckets.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import config, ops


def poly_lr_factor(epoch, epochs, power, base_lr, min_lr):
    """main.py:294-297"""
    return ((base_lr - min_lr) * (1 - epoch / epochs) ** power + min_lr) / base_lr


class FlatParams:
    """Re-homes every trainable parameter (and its .grad) into one contiguous fp32 buffer each.
    4-D conv weights keep their channels_last memory order."""
    ALIGN = 64

    def __init__(self, model, device):
        self.params = [p for p in model.parameters() if p.requires_grad]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total, self.offsets = total, offs
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad_views = []
        for p, o in zip(self.params, offs):
            pv, gv = self._view(self.flat, p, o), self._view(self.grad, p, o)
            pv.copy_(p.data)
            p.data = pv
            p.grad = gv
            self.grad_views.append(gv)

    @staticmethod
    def _view(buf, p, off):
        chunk = buf[off:off + p.numel()]
        if p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous():
            co, ci, r, s = p.shape
            return chunk.view(co, r, s, ci).permute(0, 3, 1, 2)
        return chunk.view(p.shape)

    def group_ranges(self, model):
        """Contiguous [start, end) ranges of the flat buffer per backward group, or None if the parameter order
        does not split that way (then the trainer reduces the whole buffer at the end)."""
        names = {id(p): n for n, p in model.named_parameters()}
        ranges, order = {}, []
        for p, o in zip(self.params, self.offsets):
            n = names[id(p)]
            m = n.split(".")
            g = (m[1] if m[1].startswith("layer") else "stem") if m[0] == "backbone2" else "heads"
            end = o + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if g not in ranges:
                ranges[g] = [o, end]
                order.append(g)
            elif ranges[g][1] == o:
                ranges[g][1] = end
            else:
                return None
        return {g: tuple(r) for g, r in ranges.items()}

    def bind_grads(self):
        for p, gv in zip(self.params, self.grad_views):
            p.grad = gv

    def zero_grad(self):
        ops.fill(self.grad, 0.0)
        self.bind_grads()


class WeightPrep:
    """The conv layers' derived weight operands -- ranges and fp16 pair planes (fp32 step) or bf16 copies (bf16 step), in
    the forward and the transposed (data-gradient) reading orders -- for ALL trainable conv weights in three / two launches
    right after the optimizer step (`rcf_conv_weights_prepare_{f32,bf16}`), written into persistent buffers and handed to
    the layers through their operand caches (`layers.Conv2d._derived`).  Without it every layer re-derives its operands on
    first use after each weight update: ~60 layers x 4 five-microsecond launches, 1-2 % of a training step.  Byte-identical
    to the per-layer path (tests/test_pipeline_gpu.py)."""

    ENTRY = np.dtype([("w", "<u8"), ("out", "<u8"), ("amax", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("RS", "<i4"),
                      ("first_block", "<i4"), ("nblocks", "<i4"), ("flags", "<i4"), ("pad", "<i4", (2,))])

    def __init__(self, model, precision, device):
        from . import layers, _lib
        lib = _lib.load()
        self.bf16 = precision in ("bf16", "fp16")       # 16-bit operand copies (of the library build that stores that type)
        self.half = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(precision)
        self.device = device
        self.convs = []
        for m in model.modules():
            if not isinstance(m, layers.Conv2d) or not m.weight.requires_grad or not m.weight.is_cuda:
                continue
            if m.cin_pad != m.cin or m.cout_pad != m.cout:
                continue                                    # padded copies are rebuilt per call, never cached
            if self.bf16 and (m.cin % 8 or m.cout % 8 or getattr(m, "out_fp32", False)):
                continue                                    # fp32-output heads / the stem keep their own paths
            self.convs.append(m)
        n = len(self.convs)
        self.n = n
        if n == 0:
            return
        assert self.ENTRY.itemsize == 56
        u8 = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.amax = torch.zeros(n, dtype=torch.int32, device=device)
        self.out_f, self.out_t = [], []
        kinds = ("fwd", "t") if self.bf16 else ("absmax", "fwd", "t")
        tabs = {k: np.zeros(n, dtype=self.ENTRY) for k in kinds}
        first = {k: 0 for k in kinds}
        for i, m in enumerate(self.convs):
            Cout, Cin, R, S = m.weight.shape
            w = ops.weight_rsck(m.weight)                   # raises unless the weight is channels_last ([Cout][R][S][Cin])
            elems = Cout * Cin * R * S
            if self.bf16:
                of = u8(lib.rcf_conv_weight_bf16_bytes(Cout, Cin, R, S, 0))
                ot = u8(lib.rcf_conv_weight_bf16_bytes(Cout, Cin, R, S, 1))
                blocks = {"fwd": min((of.numel() // 2 + 1023) // 1024, 256), "t": min((ot.numel() // 2 + 1023) // 1024, 256)}
                flags = {"fwd": 0, "t": 0}
            else:
                of = u8(lib.rcf_conv_weight_pairs2_bytes(Cout, Cin, R, S, 0))
                ot = u8(lib.rcf_conv_weight_pairs2_bytes(Cout, Cin, R, S, 1))
                blocks = {"absmax": max(1, min((elems // 4 + 1023) // 1024, 64)), "fwd": max(1, min((elems + 1023) // 1024, 256)),
                          "t": ((Cin + 31) // 32) * ((Cout + 31) // 32) * R * S}
                flags = {"absmax": 0, "fwd": lib.rcf_conv_pairs2_useful(Cout, Cin, R, S, 0), "t": lib.rcf_conv_pairs2_useful(Cout, Cin, R, S, 1)}
            self.out_f.append(of)
            self.out_t.append(ot)
            for k in kinds:
                e = tabs[k][i]
                e["w"], e["amax"] = w.data_ptr(), self.amax.data_ptr() + 4 * i
                e["out"] = of.data_ptr() if k == "fwd" else (ot.data_ptr() if k == "t" else 0)
                e["Cout"], e["Cin"], e["RS"] = Cout, Cin, R * S
                e["first_block"], e["nblocks"], e["flags"] = first[k], blocks[k], flags[k]
                first[k] += blocks[k]
        self.blocks = first
        self.tabs = {k: torch.from_numpy(tabs[k].view(np.uint8).reshape(-1).copy()).to(device) for k in kinds}
        self.ptrs = [ops.weight_rsck(m.weight).data_ptr() for m in self.convs]

    def run(self):
        """rebuild every derived operand from the current weights and install them in the layers' caches"""
        from . import layers
        if self.n == 0 or not layers.SCHED.cache_weight_operands:
            return
        if [ops.weight_rsck(m.weight).data_ptr() for m in self.convs] != self.ptrs:
            return                                          # the parameters moved (e.g. .to()): the per-layer path takes over
        st, p = ops._stream(), ops._p
        if self.bf16:
            _lib_call("rcf_conv_weights_prepare_bf16", p(self.tabs["fwd"]), self.blocks["fwd"], p(self.tabs["t"]), self.blocks["t"],
                      self.n, ops.CONV_FLAGS, st)
        else:
            _lib_call("rcf_conv_weights_prepare_f32", p(self.tabs["absmax"]), self.blocks["absmax"], p(self.tabs["fwd"]),
                      self.blocks["fwd"], p(self.tabs["t"]), self.blocks["t"], self.n, p(self.amax), ops.CONV_FLAGS, st)
        for i, m in enumerate(self.convs):
            key = ops.weight_key(m.weight)
            chk = ops.weight_checksum(m.weight) if ops.DEBUG_WEIGHT_CACHE else None
            cache = m.__dict__.setdefault("_wcache", {})
            if self.bf16:
                k16 = "f16" if self.half == torch.float16 else "bf16"         # layers.Conv2d._derived's key of this 16-bit type
                cache[k16], cache[k16 + "_t"] = (key, self.out_f[i], chk), (key, self.out_t[i], chk)
            else:
                cache["amax"] = (key, self.amax[i:i + 1], chk)
                cache["pairs"], cache["pairs_t"] = (key, self.out_f[i], chk), (key, self.out_t[i], chk)


def _lib_call(name, *args):
    from . import _lib
    return _lib.call(name, *args)


_GRAD_GROUPS = {}          # (SCHED.grad_group, device, world) -> (process group or None, what it is): see Trainer._make_grad_group


def make_grad_group(want, device):
    """-> (process group or None, what it is): the communicator of the gradient chunks (SCHED.grad_group).  Collective: every
    rank calls it at the same point and takes the same branches.  Two steps, each followed by an agreement (MIN of an ok flag
    over the default group, so no rank goes on to wait on a group its peers gave up on): (1) new_group; (2) a probe
    all-reduce, which makes RCCL create the communicator HERE (new_group alone is lazy), not inside the first backward.
    If any rank failed, "auto" shares the default communicator on all ranks and True raises on all ranks."""
    if want is False or want == 0:
        return None, "shared with SyncBN (SCHED.grad_group = False)"
    device = torch.device(device)

    def agree(ok):
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag) == 1
    err, g = "", None
    try:
        g = dist.new_group()
    except Exception as e:                                  # noqa: BLE001 -- reported through the mode string, decided collectively
        err = f"{type(e).__name__}: {e}"[:160]
    good = agree(not err)
    if good:
        try:
            probe = torch.ones(1, dtype=torch.float32, device=device)
            dist.all_reduce(probe, op=dist.ReduceOp.SUM, group=g)
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            if float(probe) != float(dist.get_world_size()):
                err = "the probe all-reduce returned a wrong sum"
        except Exception as e:                              # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:160]
        good = agree(not err)
    if good:
        return g, "own communicator"
    if want is True or want == 1:
        raise RuntimeError(f"SCHED.grad_group = True but the gradient communicator could not be created ({err or 'on another rank'})")
    return None, f"shared with SyncBN (own communicator failed: {err or 'on another rank'})"


class LossScaler:
    """torch.cuda.amp.GradScaler's policy (what Lightning's `precision: 16` wraps around main.py:158-178): the loss is multiplied
    by `scale` before backward; a step whose gradients hold an inf / NaN is skipped and halves the scale, `growth_interval`
    clean steps in a row double it.  Needed with fp16 storage only (bf16 has fp32's exponent range)."""

    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self.good_steps, self.skipped = 0, 0

    def update(self, finite):
        if finite:
            self.good_steps += 1
            if self.good_steps >= self.growth_interval:
                self.scale, self.good_steps = self.scale * self.growth_factor, 0
        else:
            self.scale, self.good_steps, self.skipped = self.scale * self.backoff_factor, 0, self.skipped + 1


class Trainer:
    def __init__(self, model, lr=1e-4, weight_decay=1e-4, epochs=200, power=0.9, min_lr=1e-6, device="cuda:0",
                 betas=(0.9, 0.999), eps=1e-8, force_group=False, precision=None, loss_scaler=None):
        self.model = model.to(device)
        if precision is not None:           # "bf16": bf16 activations / operands, fp32 master weights and gradients
            self.model.precision = precision
        self.device = torch.device(device)
        self.fp = FlatParams(self.model, self.device)
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.base_lr, self.weight_decay, self.epochs, self.power, self.min_lr = lr, weight_decay, epochs, power, min_lr
        self.betas, self.eps = betas, eps
        self.epoch, self.step_count = 0, 0
        # precision "fp16" (IEEE half storage: Lightning `precision: 16`) trains with a loss scaler, as the reference's GradScaler
        self.scaler = loss_scaler if loss_scaler is not None else (LossScaler() if precision == "fp16" else None)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_group: run the chunked asynchronous all-reduce path even with one rank (a dry run of the RCCL code path
        # on a single GPU: tests/test_dist_gpu.py)
        self.chunked = self.world > 1 or (force_group and dist.is_available() and dist.is_initialized())
        self.ranges = self.fp.group_ranges(self.model) if self.chunked else None
        # The gradient chunks travel on their OWN communicator where one can be had: with one process group the asynchronous
        # 10-60 MB chunk all-reduces and the small synchronous SyncBN all-reduces of the layers still in backward share one
        # RCCL stream in issue order, and every statistics exchange queues behind the chunk before it.
        # SCHED.grad_group: "auto" (default) tries the own communicator and falls back to the shared one, True insists, False shares.
        self.grad_group, self.grad_group_mode = (self._make_grad_group(config.SCHED.grad_group) if self.chunked
                                                 else (None, "none (one rank)"))
        self.profile, self.exposed = False, []             # profile: events around the wait for the gradient chunks
        if self.chunked and hasattr(self.model, "make_teacher_group"):
            self.model.make_teacher_group()                 # stage 2.1, SCHED.teacher_group: created here, collectively
        self._pending, self._done = [], set()
        self.prep = None                                    # WeightPrep, built after the first optimizer step
        if self.ranges is not None and hasattr(self.model, "grad_ready_hook"):
            self.model.grad_ready_hook = self._grads_ready

    def _make_grad_group(self, want):
        # ONE gradient communicator per process: a second Trainer (bench.py builds one per precision) reuses it instead of
        # leaving another RCCL communicator alive.  Collective all the same: every rank's Trainers are built in the same order.
        key = (str(want), str(self.device), dist.get_world_size(), id(dist.distributed_c10d._get_default_group()))   # (this init_process_group's)
        if key not in _GRAD_GROUPS:
            _GRAD_GROUPS[key] = make_grad_group(want, self.device)
        return _GRAD_GROUPS[key]

    def exposed_ms(self):
        """profile mode: per step, how long the compute stream stood waiting for the gradient chunks' all-reduces (ms)"""
        torch.cuda.synchronize(self.device)
        return [a.elapsed_time(b) for a, b in self.exposed]

    def _grads_ready(self, group):
        """tape mark: the group's gradients are final -> start their all-reduce behind the kernels queued so far"""
        r = self.ranges.get(group) if self.ranges else None
        if r is None or group in self._done:
            return
        self._done.add(group)
        self._pending.append(dist.all_reduce(self.fp.grad[r[0]:r[1]], op=dist.ReduceOp.SUM, group=self.grad_group,
                                             async_op=True))

    def lr(self):
        return self.base_lr * poly_lr_factor(self.epoch, self.epochs, self.power, self.base_lr, self.min_lr)

    def step(self, batch, check_nan=False):
        """forward + backward + gradient all-reduce + Adam.  Returns the loss dict (device tensors: no
        host sync unless check_nan, unlike main.py:167-177's per-key .item())."""
        self.model.train()
        self.fp.zero_grad()
        with torch.no_grad():
            losses = self.model(batch)
        self._pending, self._done = [], set()
        scale = self.scaler.scale if self.scaler is not None else 1.0
        self.model.run_backward(scale if self.scaler is not None else None)
        if self.chunked:
            if self.ranges is None or not hasattr(self.model, "grad_ready_hook"):
                dist.all_reduce(self.fp.grad, op=dist.ReduceOp.SUM, group=self.grad_group)
            else:
                for g in self.ranges:                           # groups whose mark did not fire (e.g. a frozen path)
                    self._grads_ready(g)
                if self.profile:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                for h in self._pending:
                    h.wait()
                if self.profile:
                    e1.record()
                    self.exposed.append((e0, e1))
        if self.scaler is not None:
            # GradScaler.step: unscale (inside Adam's grad_scale), skip the update when any gradient is not finite.  One host
            # read per step (the range of the flat gradient buffer: NaN / inf have the largest bit patterns)
            finite = (int(ops.absmax(self.fp.grad)) & 0x7FFFFFFF) < 0x7F800000
            self.scaler.update(finite)
            if not finite:
                return losses
        self.step_count += 1
        ops.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.lr(), self.step_count,
                      self.betas, self.eps, self.weight_decay, grad_scale=1.0 / (self.world * scale))
        if self.prep is None and config.SCHED.bulk_weight_prep:
            prec = self.model._select_precision()
            self.prep = WeightPrep(self.model, prec, self.device)
        if self.prep is not None:
            with ops.half_storage(self.prep.half):
                self.prep.run()                             # the next forward finds every derived weight operand ready
        if check_nan and math.isnan(float(losses["loss"])):
            raise Exception("loss is NaN")                      # main.py:176-177
        return losses
