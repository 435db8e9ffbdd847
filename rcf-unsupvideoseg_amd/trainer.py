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

import torch
import torch.distributed as dist

from . import ops


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


class Trainer:
    def __init__(self, model, lr=1e-4, weight_decay=1e-4, epochs=200, power=0.9, min_lr=1e-6, device="cuda:0",
                 betas=(0.9, 0.999), eps=1e-8, force_group=False, precision=None):
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
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_group: run the chunked asynchronous all-reduce path even with one rank (a dry run of the RCCL code path
        # on a single GPU: tests/test_dist_gpu.py)
        self.chunked = self.world > 1 or (force_group and dist.is_available() and dist.is_initialized())
        self.ranges = self.fp.group_ranges(self.model) if self.chunked else None
        # The gradient chunks travel on their OWN communicator: with one process group the asynchronous 10-60 MB chunk
        # all-reduces and the small synchronous SyncBN all-reduces of the layers still in backward would share one
        # RCCL stream in issue order, and every statistics exchange would queue behind the chunk before it.
        # RCF_GRAD_GROUP=0: the chunks share the default group (fallback: two RCCL communicators in flight at once is the
        # configuration no multi-GPU box has exercised yet)
        own_group = __import__("os").environ.get("RCF_GRAD_GROUP", "1") != "0"
        self.grad_group = dist.new_group() if (self.chunked and own_group) else None
        self._pending, self._done = [], set()
        if self.ranges is not None and hasattr(self.model, "grad_ready_hook"):
            self.model.grad_ready_hook = self._grads_ready

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
        self.model.run_backward()
        if self.chunked:
            if self.ranges is None or not hasattr(self.model, "grad_ready_hook"):
                dist.all_reduce(self.fp.grad, op=dist.ReduceOp.SUM, group=self.grad_group)
            else:
                for g in self.ranges:                           # groups whose mark did not fire (e.g. a frozen path)
                    self._grads_ready(g)
                for h in self._pending:
                    h.wait()
        self.step_count += 1
        ops.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.lr(), self.step_count,
                      self.betas, self.eps, self.weight_decay, grad_scale=1.0 / self.world)
        if check_nan and math.isnan(float(losses["loss"])):
            raise Exception("loss is NaN")                      # main.py:176-177
        return losses
