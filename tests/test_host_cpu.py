"""CPU: host-side logic of the product (config loader, synthetic data, state-dict schema, flat
parameter buffers, model construction contract)."""
import copy
import os
import types

import numpy as np
import pytest
import torch

import rcf_amd
import rcf_torch as orc
from rcf_amd import config, synth
from rcf_amd.trainer import FlatParams

REF_CFG = "/root/reference/configs/rcf/rcf_stage1.yaml"


def _args():
    return types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None)


def test_yaml_loader_base_config_and_opts(tmp_path):
    base = tmp_path / "base.yaml"
    base.write_text("a: 1\nb:\n  c: 2.5\n  d: [1, 2]\nflag: true\n")
    child = tmp_path / "child.yaml"
    child.write_text("base_config: base.yaml\nb:\n  d: [9]\nname: x\n")
    cfg = config.load_args(str(child), ["a", "7", "b.c", "0.5", "flag", "false"])
    assert cfg.a == 7 and cfg.b == {"c": 0.5, "d": [9]} and cfg.flag is False and cfg.name == "x"
    dup = tmp_path / "dup.yaml"
    dup.write_text("a: 1\na: 2\n")
    with pytest.raises(ValueError):
        config.load_yaml(str(dup))
    with pytest.raises(KeyError):
        config.load_args(str(child), ["missing.key", "1"])


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="reference configs only exist in the build container")
def test_reference_stage1_yaml_builds_the_model():
    """the reference's own YAML drives the constructor unchanged (mask size aside)"""
    args = config.load_args(REF_CFG, ["model_kwargs.mask_size", "[24, 40]"]) if False else config.load_args(REF_CFG)
    kw = copy.deepcopy(args.model_kwargs)
    want = config.stage1_model_kwargs((96, 96), dropout=0.1, norm="SyncBN")
    assert kw == want, "stage1_model_kwargs drifted from configs/rcf/rcf_stage1.yaml"
    args.object_channel = None
    m = rcf_amd.RCFModel(args, **kw)
    assert "type" not in kw["backbone2"]              # popped like the reference does (one model per dict)
    assert m.num_classes == 4 and m.mask_size == (96, 96)


def test_state_dict_schema_equals_oracle_and_counts():
    kw = config.stage1_model_kwargs((24, 40), dropout=0.0)
    hip, ora = rcf_amd.RCFModel(_args(), **copy.deepcopy(kw)), orc.RCFModel(_args(), **copy.deepcopy(kw))
    a, b = hip.state_dict(), ora.state_dict()
    assert list(a) == list(b) and len(a) == 354
    assert all(tuple(a[k].shape) == tuple(b[k].shape) and a[k].dtype == b[k].dtype for k in a)
    assert sum(p.numel() for p in hip.parameters()) == 39482902
    # a plain (contiguous) checkpoint loads into the channels_last parameters and round-trips
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in a.items()}).items()}
    hip.load_state_dict(sd)
    back = hip.state_dict()
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    w = hip.backbone2.layer1[0].conv2.weight if hasattr(hip.backbone2.layer1, "__getitem__") else \
        getattr(hip.backbone2.layer1, "0").conv2.weight
    assert w.permute(0, 2, 3, 1).is_contiguous()
    # stage 2.1 adds the EMA copies under the reference's names
    kw2 = config.stage1_model_kwargs((24, 40), dropout=0.0)
    kw2["backbone2"]["create_ema"] = True
    kw2["decode_head2"]["create_ema"] = True
    m2 = rcf_amd.RCFModel(_args(), **kw2)
    keys = set(m2.state_dict())
    assert any(k.startswith("backbone2_ema.") for k in keys) and any(k.startswith("decode_head2_ema.") for k in keys)
    assert not any(p.requires_grad for p in m2.backbone2_ema.parameters())


def test_flat_params_views_and_zero_init():
    kw = config.stage1_model_kwargs((24, 40), dropout=0.0)
    m = rcf_amd.RCFModel(_args(), **kw)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    fp = FlatParams(m, torch.device("cpu"))
    assert fp.total >= 39482902 and fp.total % FlatParams.ALIGN == 0
    for (n, p), o in zip(((n, p) for n, p in m.named_parameters() if p.requires_grad), fp.offsets):
        assert torch.equal(p.detach(), before[n]), n                          # values preserved
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * o and o % 4 == 0      # 16-byte aligned views
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o
    from rcf_amd.layers import Conv2d
    for mod in m.modules():                                # the HIP convs keep [Cout][R][S][Cin] memory order
        if isinstance(mod, Conv2d) and mod.cin % 4 == 0 and mod.weight.requires_grad:
            assert mod.weight.permute(0, 2, 3, 1).is_contiguous()
            assert mod.weight.grad.permute(0, 2, 3, 1).is_contiguous()


def test_synthetic_data_is_deterministic_and_well_formed():
    a, b = synth.make_batch(2, 48, 64, config_id=3), synth.make_batch(2, 48, 64, config_id=3)
    for k in ("imgs", "gt_fw_flows", "gt_bw_flows"):
        for x, y in zip(a[k], b[k]):
            assert np.array_equal(x, y)
    assert a["imgs"][0].shape == (2, 3, 48, 64) and a["gt_fw_flows"][0].shape == (2, 2, 48, 64)
    assert np.sqrt((a["gt_fw_flows"][0] ** 2).sum(1)).max() <= 20.0 + 1e-4
    assert synth.smooth_rgb(16, 16, 1).dtype == np.uint8 and not np.array_equal(synth.smooth_rgb(16, 16, 1), synth.smooth_rgb(16, 16, 2))
    assert config.mask_size_for(480, 854) == (120, 214) and config.mask_size_for(384, 384) == (96, 96)


def test_unsupported_options_raise_instead_of_silently_diverging():
    kw = config.stage1_model_kwargs((24, 40))
    kw["decode_head3"]["create_flownet"] = True
    with pytest.raises(NotImplementedError):
        rcf_amd.RCFModel(_args(), **kw)
    kw = config.stage1_model_kwargs((24, 40))
    kw["decode_head"]["free_residual"] = False
    with pytest.raises(NotImplementedError):
        rcf_amd.RCFModel(_args(), **kw)
    m = rcf_amd.RCFModel(_args(), **config.stage1_model_kwargs((24, 40)))
    batch = {k: [torch.zeros(1, c, 96, 160)] * n for k, c, n in (("imgs", 3, 2), ("gt_fw_flows", 2, 1), ("gt_bw_flows", 2, 1))}
    m.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(batch)


def test_export_grid_and_rounding():
    """export.py (stand-in for torchvision.utils.save_image, absent here: unpinned against it): single image passes
    through, a batch is tiled with 2-pixel padding, values are rounded as x*255+0.5"""
    import torch
    from rcf_amd import export
    one = torch.rand(1, 3, 5, 7)
    assert torch.equal(export.make_grid(one), one[0])
    assert tuple(export.make_grid(torch.rand(5, 7)).shape) == (3, 5, 7)          # grey -> 3 channels
    g = export.make_grid(torch.ones(10, 3, 5, 7))                                # 8 per row, 2 rows
    assert tuple(g.shape) == (3, 2 * 7 + 2, 8 * 9 + 2)
    assert float(g[:, :2].abs().max()) == 0 and float(g[:, 2:7, 2:9].min()) == 1 and float(g[:, 9:14, 20:].max()) == 0
    u8 = export.to_uint8_hwc(torch.tensor([[[0.0, 0.5, 1.0, 1.2, -0.3, 0.998]]]))
    assert u8[0, :, 0].tolist() == [0, 128, 255, 255, 0, 254] and u8.shape == (1, 6, 3)


def test_vit_state_dict_schema_matches_reference(golden_dir):
    """parameter names / shapes of rcf_amd.vit.vit_small(patch_size=8) == models/dino_vit.py's (captured in the fixture)"""
    import os
    import numpy as np
    from rcf_amd import vit
    fx = np.load(os.path.join(golden_dir, "vit_small8.npz"))
    mine = [f"{k}:{'x'.join(map(str, v.shape))}" for k, v in vit.vit_small(patch_size=8).state_dict().items()]
    assert mine == [str(s) for s in fx["schema"]]


def test_bench_line_helpers():
    """bench.py's host-side pieces that need no GPU: strict JSON (a diverged loss must not print NaN), the bracket families
    the default headline names exist, flags parse"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("rcf_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    dirty = {"loss": float("nan"), "x": [1.0, float("inf"), {"y": -float("inf"), "z": 2}], "s": "ok", "n": None, "i": 3}
    clean = b._finite(dirty)
    assert clean == {"loss": None, "x": [1.0, None, {"y": None, "z": 2}], "s": "ok", "n": None, "i": 3}
    json.loads(json.dumps(clean, allow_nan=False))
    assert "conv_dgrad_wide" in b.FAMILIES_F32 and "conv_wgrad_h2t4" in b.FAMILIES_F32 and "conv_bf16_wgrad4" in b.FAMILIES_BF16
    assert b.HBM_PEAK_GBS == 8000.0 and b.BF16_MFMA_PEAK_TF == 2500.0


def test_bench_contract_line_is_short_and_complete(capsys, tmp_path, monkeypatch):
    """The driver keeps an 8 KB tail of stdout and parses the LAST line: round 4's 20.7 KB line came back `parsed: null`.
    The contract line built from a full round-4 result (the largest this repo has printed) must stay under 2 KB, carry every
    contract key incl. `roofline` and `cpu_baseline`, and be the last thing `emit` prints."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rcf_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = json.load(open(os.path.join(root, "profiles", "r04_bench_n1.json")))
    assert len(json.dumps(full)) > 16000
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    b.emit(full)
    lines = capsys.readouterr().out.strip().split("\n")
    assert lines[-2].startswith("BENCH_DETAIL ") and json.loads(lines[-2][len("BENCH_DETAIL "):]) == full
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json")) == full
    assert len(lines[-1]) <= b.CONTRACT_LINE_MAX <= 2048
    line = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "bf16_frames_per_s", "bf16_ms_per_step", "crf_ms_per_frame",
              "crf_frac", "warp_frac"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and len(line["config"]["workload"]) <= 120 and len(line["roofline"]["kernel"]) <= 80
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    # a worst-case result (every string at its clip length) still fits
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "w" * 500
    fat["roofline"]["kernel"] = "k" * 500
    fat["cpu_baseline"]["sample"] = "s" * 500
    assert len(json.dumps(b.contract_line(fat))) <= b.CONTRACT_LINE_MAX
    # ... and a line that would NOT fit is degraded, never refused: optional keys go, then the strings shrink;
    # the contract's own keys stay and the size is counted in encoded bytes
    huge = b.contract_line(fat)
    huge["config"]["workload"] = "\u00e9" * 1500                        # 2 bytes each in UTF-8, 6 as JSON escapes
    huge["roofline"]["kernel"] = "k" * 900
    huge.update(comm_note="x" * 10)
    text = b.fit_contract_line(huge)
    small = json.loads(text)
    assert len(text.encode()) <= b.CONTRACT_LINE_MAX
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in small, k
    assert small["value"] == full["value"] and small["roofline"]["frac"] == full["roofline"]["frac"] and "detail" not in small
    # N > 1: the communication keys a scaling run is read against travel on the line
    multi = json.loads(json.dumps(full))
    multi["comm"] = {"syncbn_collectives": 106, "syncbn_ms_per_step": 3.2, "allreduce_exposed_ms": 0.4,
                     "grad_communicator": "own communicator"}
    ml = b.contract_line(multi)
    assert ml["syncbn_collectives"] == 106 and ml["allreduce_exposed_ms"] == 0.4 and ml["grad_comm"] == "own communicator"
    assert len(b.fit_contract_line(ml).encode()) <= b.CONTRACT_LINE_MAX and "syncbn_ms_per_step" in json.loads(b.fit_contract_line(ml))


def test_weight_operand_cache_invalidation(monkeypatch):
    """operands derived from a conv weight are cached by (data_ptr, _version, epoch): an in-place write on the parameter
    bumps `_version`, a write through `.data` bumps nothing -- the library's own `.data` writers (copy_param_and_buffer,
    the EMA update, model.train() / load_state_dict) bump the epoch, and the debug mode catches a writer that does not"""
    from rcf_amd import layers, ops
    conv = layers.Conv2d(4, 8, 3)
    made = []

    def make():
        made.append(float(conv.weight.detach().abs().max()))
        return made[-1]
    assert conv._derived("amax", make) == conv._derived("amax", make) and len(made) == 1      # hit
    with torch.no_grad():
        conv.weight.mul_(2.0)                           # torch in-place op on the parameter: _version changes
    conv._derived("amax", make)
    assert len(made) == 2
    conv.weight.data.mul_(2.0)                          # .data write: invisible to the key ...
    assert conv._derived("amax", make) == made[1] and len(made) == 2
    ops.weights_changed()                               # ... until the writer says so
    assert conv._derived("amax", make) == 2 * made[1] and len(made) == 3
    # the repo's own .data writers bump the epoch
    src, dst = layers.Conv2d(4, 8, 3), layers.Conv2d(4, 8, 3)
    e0 = ops.WEIGHT_EPOCH[0]
    rcf_amd.model.copy_param_and_buffer(src, dst)
    assert ops.WEIGHT_EPOCH[0] > e0 and torch.equal(src.weight, dst.weight)
    kw = config.stage1_model_kwargs((24, 40), dropout=0.0, norm="BN")
    m = rcf_amd.RCFModel(types.SimpleNamespace(checkpoints_dir="/tmp/rcf_t", object_channel=None), **kw)
    m.eval()
    for fn in (lambda: m.train(), lambda: m.eval(), lambda: m.load_state_dict(m.state_dict())):      # transitions / loads
        e0 = ops.WEIGHT_EPOCH[0]
        fn()
        assert ops.WEIGHT_EPOCH[0] > e0
    e0 = ops.WEIGHT_EPOCH[0]
    m.eval()                                            # no transition: a trainer calling train() every step keeps its caches
    assert ops.WEIGHT_EPOCH[0] == e0
    # the EMA update writes the TEACHER only: the student's cached operands (prepared in bulk after the optimizer step) survive it
    student, teacher = layers.Conv2d(4, 8, 3), layers.Conv2d(4, 8, 3)
    student._derived("amax", lambda: 1.0)
    teacher._derived("amax", lambda: 2.0)
    key0, e0 = student.__dict__["_wcache"]["amax"][0], ops.WEIGHT_EPOCH[0]
    rcf_amd.model.momentum_update_param_and_buffer(student, teacher, 0.9)
    assert ops.WEIGHT_EPOCH[0] == e0 and student.__dict__["_wcache"]["amax"][0] == key0 == ops.weight_key(student.weight)
    assert "_wcache" not in teacher.__dict__
    # debug mode: a stale hit raises instead of being used
    monkeypatch.setattr(ops, "DEBUG_WEIGHT_CACHE", True)
    conv2 = layers.Conv2d(4, 8, 3)
    conv2._derived("amax", lambda: 1)
    conv2.weight.data.mul_(3.0)
    with pytest.raises(ops._lib.RcfHipError):
        conv2._derived("amax", lambda: 1)


def test_grad_sketch_estimates_vector_distance_and_matches_the_committed_fixture_format(golden_dir):
    """synth.grad_sketch / sketch_error (the fingerprints behind test_fullsize_b8_gradients_vs_oracle): deterministic, linear,
    an estimator of the relative vector distance good to ~10 % with a few hundred numbers, and the committed float64 fixture
    carries one k-vector per parameter tensor of the model"""
    import json
    g = torch.Generator().manual_seed(0)
    truth = {f"m.{i}.w": torch.randn(257, 33, generator=g, dtype=torch.float64) * (1 + i) for i in range(12)}
    noisy = {k: v + 3e-3 * v.abs().mean() * torch.randn(v.shape, generator=g, dtype=torch.float64) for k, v in truth.items()}
    st, sn = synth.grad_sketch(truth, k=32), synth.grad_sketch(noisy, k=32)
    assert st == synth.grad_sketch({k: v.clone() for k, v in truth.items()}, k=32)                      # deterministic
    both = synth.grad_sketch({k: truth[k] + noisy[k] for k in truth}, k=32)
    assert all(abs(a + b - c) <= 1e-9 * (abs(a) + abs(b) + 1) for n in st for a, b, c in zip(st[n], sn[n], both[n]))   # linear
    true_rel = (sum(float(((noisy[k] - truth[k]) ** 2).sum()) for k in truth) / sum(float((truth[k] ** 2).sum()) for k in truth)) ** 0.5
    est = synth.sketch_error(sn, st, "m.")
    assert abs(est / true_rel - 1) < 0.15, (est, true_rel)
    assert synth.sketch_error(st, st) == 0.0 and synth.sketch_error(sn, st, "absent.") == 0.0
    fx = json.load(open(os.path.join(golden_dir, "oracle_b8_selfdev.json")))
    m = rcf_amd.RCFModel(types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False),
                         **config.stage1_model_kwargs(config.mask_size_for(fx["H"], fx["W"]), dropout=0.0, affine=False, norm="BN"))
    names = {n for n, p in m.named_parameters() if p.requires_grad}
    assert set(fx["sketch_f64"]) <= names and all(len(v) == fx["sketch_k"] for v in fx["sketch_f64"].values())
    assert set(fx["vector_fp32_err"]) == set(fx["gradnorm_f64"]) == {n.split(".")[0] for n in fx["sketch_f64"]}


def test_schedule_object_is_the_only_home_of_the_step_knobs(monkeypatch):
    """rcf_amd.config.Schedule: documented fields, explicit setters, no environment feed; the package's modules read
    config.SCHED at call time, and the only environment variables the package still reads are the two documented ones"""
    import glob
    import re
    from rcf_amd import config, layers, ops
    assert layers.SCHED is config.SCHED and ops.SCHED is config.SCHED
    d = config.Schedule().as_dict()
    assert d["overlap_wgrad"] and d["late_wgrad"] and d["planes"] and d["fold_bn"] and not d["fuse_bn_bwd"]
    assert d["grad_group"] == "auto" and d["teacher_group"] is False         # own gradient communicator where one can be had
    for k in d:
        assert re.search(r"\b" + k + r"\b", config.Schedule.__doc__), f"Schedule.{k} is not documented"
    sc = config.Schedule()
    old = sc.set(fold_bn=False, join_planes="all")
    assert old == {"fold_bn": True, "join_planes": "stage"} and sc.fold_bn is False and sc.join_planes == "all"
    sc.parse(["fold_bn=1", "side_priority=-1", "h2_kinds=fd", "overlap_wgrad=0"])
    assert sc.fold_bn is True and sc.side_priority == -1 and sc.h2_kinds == "fd" and sc.overlap_wgrad is False
    with pytest.raises(AttributeError):
        sc.set(no_such_knob=1)
    monkeypatch.setenv("RCF_FOLD_BN", "0")
    assert config.Schedule().fold_bn is True                                  # the environment does not feed it
    pkg = os.path.dirname(os.path.abspath(rcf_amd.__file__))
    envs = set()
    for f in glob.glob(os.path.join(pkg, "*.py")):
        if os.path.basename(f) == "build.py":
            continue
        envs |= set(re.findall(r"environ(?:\.get)?[\[(]\s*\"(RCF_[A-Z0-9_]+)\"", open(f).read()))
    assert envs <= {"RCF_CONV_FLAGS", "RCF_DEBUG_WEIGHT_CACHE"}, envs
