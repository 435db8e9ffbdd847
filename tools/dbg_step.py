import os, sys, time, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, ops, layers
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 480, 854
args = types.SimpleNamespace(checkpoints_dir="/tmp/x", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device="cuda:0")
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
batch = {"imgs": [t(x) for x in nb["imgs"]], "gt_fw_flows": [t(x) for x in nb["gt_fw_flows"]], "gt_bw_flows": [t(x) for x in nb["gt_bw_flows"]]}
orig = layers.commuted_concat_conv
import rcf_amd.backbone as bb
def traced(a, b, conv, tape):
    torch.cuda.synchronize(); t0 = time.time(); print("commuted fwd start", flush=True)
    y = orig(a, b, conv, tape)
    torch.cuda.synchronize(); print("commuted fwd done %.3f s" % (time.time() - t0), flush=True)
    f = tape.ops[-1]
    def g():
        torch.cuda.synchronize(); t1 = time.time(); print("commuted bwd start", flush=True)
        f(); torch.cuda.synchronize(); print("commuted bwd done %.3f s" % (time.time() - t1), flush=True)
    tape.ops[-1] = g
    return y
bb.commuted_concat_conv = traced
for i in range(2):
    t0 = time.time(); l = tr.step(batch); torch.cuda.synchronize(); print("step", i, time.time() - t0, float(l["loss"]), flush=True)
