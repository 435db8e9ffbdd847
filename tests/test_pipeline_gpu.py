"""GPU: the pieces either side of the hot path composed the way a maintainer would wire them: decoded
u8 batch -> rcf_amd.data_pipeline.Transform on the device -> RCFModel / Trainer step (stage 2.2: pseudo-label loss on the
transformed masks) -> evaluation transform -> forward_eval -> Evaluator.  No parity claim of its own (each piece has its
fixtures); this checks the interfaces fit: shapes, dtypes, list-of-tensors batch layout, finite losses that move."""
import types

import numpy as np
import pytest
import torch

import rcf_amd
from rcf_amd import config, evaluate, synth
from rcf_amd.data_pipeline import Transform

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _raw_batch(seeds, H, W):
    s = [synth.loader_sample(k, H, W) for k in seeds]
    return {"imgs": torch.from_numpy(np.stack([x["frames"] for x in s])).to(DEV),
            "gt_fw_flows": torch.from_numpy(np.stack([x["fw"] for x in s])).to(DEV),
            "gt_bw_flows": torch.from_numpy(np.stack([x["bw"] for x in s])).to(DEV),
            "pl_masks": torch.from_numpy(np.stack([x["pl"] for x in s])).to(DEV),
            "seq_names": [f"synth{k}" for k in seeds]}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_transform_feeds_the_training_step_and_the_evaluator(precision, report):
    H, W = 400, 520
    tf = Transform(training=True, strong_aug=True, has_pl=True)
    raw = _raw_batch([501, 502], H, W)
    batch = tf(raw, rng=np.random.RandomState(11))
    assert [tuple(t.shape) for t in batch["imgs"]] == [(2, 3, 384, 384)] * 2 and len(batch["pl_masks"]) == 2
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_pipe", object_channel=1, eval_save=False, eval_export=False,
                                 eval_pos_th=0.35, rank=-1, set_object_channel_after_epoch=1)
    mask = config.mask_size_for(384, 384)
    model = rcf_amd.RCFModel(args, **config.stage22_model_kwargs(mask, dropout=0.0, norm="BN"))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=torch.device(DEV), precision=precision)
    losses = [tr.step(batch) for _ in range(2)]
    vals = [{k: float(v) for k, v in l.items()} for l in losses]
    assert all(np.isfinite(list(v.values())).all() for v in vals) and "loss_pl" in vals[0]
    assert vals[0]["loss"] != vals[1]["loss"], "the second step runs on updated weights"
    # evaluation: single frames, resized (ratio 0.98) and normalised on the device, masks against a 0/128/255 annotation
    etf = Transform(training=False)
    eraw = {"imgs": raw["imgs"][:, :1].contiguous()}
    ebatch = etf(eraw)
    assert tuple(ebatch["imgs"][0].shape) == (2, 3, 392, int(W * (392 / H) + 0.5))
    ann = torch.from_numpy(np.stack([np.where(synth.soft_blob_mask(H, W, 9 + i) > 0.5, 255, 0).astype(np.uint8) for i in range(2)])).to(DEV)
    ebatch.update(ann=ann, seq_names=["a", "b"])
    ev = evaluate.Evaluator(args, mask_layer=model.mask_layer if hasattr(model, "mask_layer") else 4)
    ious = ev.test_step(model, ebatch)
    assert ious.shape[0] == 2 and np.all((ious >= 0) & (ious <= 1) | np.isnan(ious))
    miou, frame_avg, per_seq = ev.test_epoch_end(current_epoch=0, testing=True)
    report(f"pipeline [{precision}]: transform -> 2 train steps (loss {vals[0]['loss']:.4f} -> {vals[1]['loss']:.4f}, loss_pl "
           f"{vals[0]['loss_pl']:.4f}) -> eval transform -> Evaluator mIoU {float(miou):.4f}")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_steps_are_bit_reproducible(precision, report):
    """two runs of three steps from the same weights and batch (Dropout2d off): identical losses, gradients and parameters,
    bit for bit -- fixed-order reductions everywhere (split-K, batch-norm sums, loss sums), no floating-point atomics"""
    H, W, B = 96, 160, 2

    def run():
        args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_rep", object_channel=None, eval_save=False, eval_export=False)
        model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="SyncBN"))
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
        tr = rcf_amd.Trainer(model, device=torch.device(DEV), precision=precision)
        nb = synth.make_batch(B, H, W, config_id=2)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
        batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
        out = []
        for _ in range(3):
            loss = float(tr.step(batch)["loss"])
            out.append((loss, tr.fp.grad.clone(), tr.fp.flat.clone()))
        return out
    a, b = run(), run()
    for (la, ga, pa), (lb, gb, pb) in zip(a, b):
        assert la == lb and torch.equal(ga, gb) and torch.equal(pa, pb)
    report(f"reproducibility [{precision}]: 3 steps twice, losses {[round(x[0], 5) for x in a]}, gradients and parameters bit-identical")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_bulk_weight_preparation_equals_the_per_layer_path(prec):
    """trainer.WeightPrep rebuilds the derived operands of every conv weight (ranges + fp16 pair planes in both reading orders,
    or the two bf16 copies) in three / two launches after the optimizer step: the layers must find, in their caches, the very
    bytes the per-layer calls produce from the same weights -- and a step with the bulk path must equal one without it"""
    import copy
    import types
    import numpy as np
    import rcf_amd
    from rcf_amd import config, layers, ops, synth, trainer
    H, W, B = 64, 96, 1
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_t", object_channel=None)
    nb = synth.make_batch(B, H, W, config_id=1)
    batch = {k: [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
    losses = {}
    for bulk in (True, False):
        rcf_amd.config.SCHED.bulk_weight_prep = bulk
        m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
        tr = rcf_amd.Trainer(m, device="cuda:0", precision=prec)
        losses[bulk] = [float(tr.step(batch)["loss"]) for _ in range(3)]
        if bulk:
            assert tr.prep is not None and tr.prep.n >= 50
            checked = 0
            for conv in tr.prep.convs:
                w = conv.weight
                key = ops.weight_key(w)
                c = conv._wcache
                if prec == "fp32":
                    assert c["amax"][0] == key and c["pairs"][0] == key and c["pairs_t"][0] == key
                    aw = ops.absmax(ops.weight_rsck(w))
                    assert int(aw) == int(c["amax"][1])
                    Cout, Cin, R, S = w.shape
                    for kind, fresh, tr_ in (("pairs", ops.weight_pairs(w, aw), 0), ("pairs_t", ops.weight_pairs_t(w, aw), 1)):
                        if (R * S * (Cout if tr_ else Cin)) % 16:
                            continue                        # K not a whole number of K-steps: the buffers hold unwritten padding
                        got = c[kind][1]
                        half = got.numel() // 2
                        assert torch.equal(got[:half], fresh[:half]), (kind, tuple(w.shape))
                        if ops._lib.load().rcf_conv_pairs2_useful(Cout, Cin, R, S, tr_):
                            assert torch.equal(got[half:], fresh[half:]), (kind, "plane-separated half", tuple(w.shape))
                else:
                    assert c["bf16"][0] == key and c["bf16_t"][0] == key
                    assert torch.equal(c["bf16"][1], ops.weight_bf16(w)) and torch.equal(c["bf16_t"][1], ops.weight_bf16(w, True))
                checked += 1
            assert checked == tr.prep.n
    rcf_amd.config.SCHED.bulk_weight_prep = True
    assert losses[True] == losses[False], losses


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_second_stream_for_weight_gradients_changes_nothing(prec):
    """the optional second HIP stream for the weight gradients (config.SCHED.overlap_wgrad) only
    reorders independent launches: losses of three steps and the parameters after them are identical with and without it"""
    import copy
    import types
    import numpy as np
    from rcf_amd import layers
    H, W, B = 64, 96, 1
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_t", object_channel=None)
    nb = synth.make_batch(B, H, W, config_id=1)
    batch = {k: [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
    saved = layers.SCHED.overlap_wgrad, layers.SCHED.late_wgrad
    res = {}
    try:
        # False / True: one stream / the weight gradient beside its layer's data gradient; "late": started after the data
        # gradient, i.e. beside the next layer's batch-norm backward (SCHED.late_wgrad; fp32 path)
        for overlap in (False, True, "late"):
            layers.SCHED.overlap_wgrad, layers.SCHED.late_wgrad = bool(overlap), overlap == "late"
            m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
            tr = rcf_amd.Trainer(m, device="cuda:0", precision=prec)
            losses = [float(tr.step(batch)["loss"]) for _ in range(3)]
            torch.cuda.synchronize()
            res[overlap] = (losses, tr.fp.flat.clone())
    finally:
        layers.SCHED.overlap_wgrad, layers.SCHED.late_wgrad = saved
    assert res[False][0] == res[True][0] == res["late"][0], (res[False][0], res[True][0], res["late"][0])
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][1], res["late"][1])
