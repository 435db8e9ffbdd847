#!/usr/bin/env python3
"""Headline benchmark: RCF stage-1 training frames/sec at 480x854 (BASELINE.json `metric`).

One step = forward + backward + gradient all-reduce + Adam on 8 pairs (16 frames) per GPU of
synthetic 480x854 RGB + flow, fp32, Dropout2d 0.1, SyncBN, random-init weights of
the reference architecture.  frames/s = 2 * pairs_per_gpu * n_gpus * steps / time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0's LAST stdout line is the contract line: one JSON object of at most CONTRACT_LINE_MAX bytes (asserted) with the
contract keys only -- `roofline` for the kernel instantiation with the largest share of the step's kernel time (the 256-wide
plane forward kernel conv_h2d_kernel<4,false,false>): algorithmic FLOPs of its launches / their HIP-event durations;
`cpu_baseline` = the oracle (CPU restatement of the reference) timed on a bounded sample.  Everything else (per-kernel tables,
the bf16 step's own roofline, CRF / warp / stage 2.1 / ViT / loader legs, prose) goes to `gpurun_out/bench_detail.json` and to
an EARLIER stdout line that starts with "BENCH_DETAIL " (so that no parser mistakes it for the contract line).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
GF_PER_FRAME = 2043.3              # fwd+bwd algorithmic GFLOP per 480x854 frame
BF16_MFMA_PEAK_TF = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA
X3_MFMA_PEAK_TF = 2500.0 / 6       # fp32 product = 6 bf16 partial products on the bf16 matrix cores (ViT GEMMs, attention)
H2_MFMA_PEAK_TF = 2500.0 / 3       # convs: fp32 product = 3 fp16 partial products (operands split into fp16 pairs)
PRIMING_STEPS = int(os.environ.get("RCF_BENCH_PRIMING", "2"))   # untimed set-up steps before the warm-up (allocator / stream scratch)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E, 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=8, help="pairs per GPU (configs/rcf/rcf_stage1.yaml:4)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--crf-iters", type=int, default=5)
    ap.add_argument("--no-stage2", action="store_true")
    ap.add_argument("--no-bf16", action="store_true")
    ap.add_argument("--no-fp16", action="store_true")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  The parent has not touched the GPU
        # (torch is not even imported yet) and never will: it relays the ranks' output -- rank 0's JSON line -- and exits
        # with the launcher's code.  The torchrun-launched form (WORLD_SIZE set) skips this.
        sys.exit(_self_launch(a.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import types

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one rank per GPU.  RCF_BENCH_BACKEND=gloo lets several ranks share one device (a functional check of the N > 1
    # path on a single-GPU box: RCCL refuses two ranks per GPU); the driver's runs use RCCL ("nccl").
    backend = os.environ.get("RCF_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import rcf_amd
    from rcf_amd import config, ops, synth

    H, W, B = a.height, a.width, a.pairs
    mask = config.mask_size_for(H, W)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False,
                                 eval_export=False)
    nb = synth.make_batch(B, H, W, config_id=2, first_index=rank * B)      # weak scaling: B pairs per rank
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    batch = {"imgs": [t(x) for x in nb["imgs"]], "gt_fw_flows": [t(x) for x in nb["gt_fw_flows"]],
             "gt_bw_flows": [t(x) for x in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"],
             "seq_names": nb["seq_names"], "paths": nb["paths"]}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step_leg(precision, family, steps, warmup):
        """One training configuration: PRIMING_STEPS + warmup untimed steps, then `steps` timed ones between barriers
        (max over ranks), `family` bracketed live inside the timed region; then two more steps with EVERY conv family
        bracketed (roofline_by_kernel)."""
        model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(mask, dropout=0.1, norm="SyncBN"))
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
        # fp16 storage trains with a loss scaler (GradScaler's policy); the bench starts it at 2^10 so that no step of the timed
        # region is a skipped one (reported as `skipped_steps`: a skipped step does no Adam / weight preparation)
        scaler = rcf_amd.trainer.LossScaler(2.0 ** 10) if precision == "fp16" else None
        trainer = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=dev, precision=precision, loss_scaler=scaler)
        # set-up, not measurement: priming steps let the caching allocator and the second stream's scratch reach their
        # steady state (the first steps hipMalloc); then the W untimed warm-up steps and the K timed ones of the contract
        for _ in range(PRIMING_STEPS + warmup):
            trainer.step(batch)
        barrier()
        skipped0 = scaler.skipped if scaler is not None else 0
        ops.PROFILE.start(family)
        t0 = time.perf_counter()
        for _ in range(steps):
            losses = trainer.step(batch)
        barrier()
        dt = time.perf_counter() - t0
        prof = ops.PROFILE.stop()
        if scaler is not None:
            step_leg.fp16_info = {"loss_scale_log2": int(round(float(np.log2(scaler.scale)))), "skipped_steps_in_timed_region": scaler.skipped - skipped0,
                                  "skipped_steps_before": skipped0}
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0])
        loss_val = float(losses["loss"])
        if loss_val != loss_val:
            raise SystemExit("loss is NaN")
        fams = FAMILIES_F32 if precision == "fp32" else FAMILIES_BF16
        ops.PROFILE.start(list(fams))
        for _ in range(2):
            trainer.step(batch)
        by = ops.PROFILE.stop()
        # ... and once more with the weight gradients on the SAME stream as everything else: with the second stream a
        # weight gradient and a data gradient share the chip and each one's bracket also counts what the other costs it
        from rcf_amd import layers
        saved = layers.SCHED.overlap_wgrad
        layers.SCHED.overlap_wgrad = False
        try:
            trainer.step(batch)
            ops.PROFILE.start(list(fams))
            for _ in range(2):
                trainer.step(batch)
            by_serial = ops.PROFILE.stop()
        finally:
            layers.SCHED.overlap_wgrad = saved
        # N > 1: what the collectives cost, so that a scaling run can be read (DistCtx.profile brackets every SyncBN exchange
        # on the stream it is issued on; Trainer.profile brackets the wait for the gradient chunks at the end of backward)
        comm = None
        if world > 1:
            d, nprof = model._dist(), 2
            d.count, d.bytes, d.events, d.profile = 0, 0, [], True
            trainer.profile, trainer.exposed = True, []
            for _ in range(nprof):
                trainer.step(batch)
            ms, ex = d.profile_ms(), trainer.exposed_ms()
            d.profile, trainer.profile = False, False
            comm = {"syncbn_collectives": d.count // nprof, "syncbn_ms_per_step": round(sum(ms) / nprof, 3),
                    "syncbn_bytes_per_step": d.bytes // nprof, "syncbn_max_us": round(1e3 * max(ms), 1) if ms else None,
                    "allreduce_exposed_ms": round(sum(ex) / max(len(ex), 1), 3), "grad_communicator": trainer.grad_group_mode,
                    "grad_bytes": int(trainer.fp.total) * 4, "grad_chunks": len(trainer.ranges or {}) or 1}
        barrier()
        del trainer, model
        torch.cuda.empty_cache()
        return dt, loss_val, prof, by, by_serial, comm

    def roofline_of(prof, name, kernel, peak, by=None, traffic=None):
        n, flops, ms = prof["launches"], prof["flops"], prof["ms"]
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"kernel": kernel, "family": name, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "launches": n,
                "avg_launch_ms": round(ms / max(n, 1), 4), "flops_per_launch": round(flops / max(n, 1), 1)}

    def by_kernel(by, fams, peak, nsteps=2):
        out = {}
        for name, kernel in fams.items():
            r = by.get(name)
            if not r or not r["launches"]:
                continue
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
            out[name] = {"kernel": kernel, "launches_per_step": r["launches"] // nsteps,
                         "ms_per_step": round(r["ms"] / nsteps, 3), "avg_launch_ms": round(r["ms"] / r["launches"], 4),
                         "flops_per_launch": round(r["flops"] / r["launches"], 1), "achieved": round(ach, 1),
                         "frac": round(ach / peak, 4)}
        return out

    # ---- headline: fp32 step (BASELINE configs[1]).  Bracketed in the timed region: the kernel instantiation with the largest
    # share of the step's kernel time -- since round 4 (pair planes; join planes only in front of stage-first blocks) that is the
    # 256-wide plane FORWARD kernel conv_h2d_kernel<4,false,false> (15.0-15.7 ms of a step, roofline_by_kernel_one_stream; the
    # same kernel on the transposed weights = the data gradient 14.1-14.9, the plane weight gradient 13.4-14.1).  The forward pass
    # has no second stream beside it, so its live bracket and its one-stream bracket measure the same thing.
    fam32 = os.environ.get("RCF_BENCH_FAMILY", "conv_h2d_fwd")
    dt, loss_val, prof, by32, by32s, comm32 = step_leg("fp32", fam32, a.steps, a.warmup)
    frames = 2 * B * world * a.steps
    value = frames / dt
    out = {
        "metric": "training frames/sec at 480x854 (RCF stage-1)", "value": round(value, 3), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (3xfp16-MFMA split)", "data": "synthetic",
        "priming_steps": PRIMING_STEPS,
        "arithmetic": "fp32 step, emulated on the fp16 matrix cores: every conv operand is scaled by a power of two into "
                      "fp16's range and split into 2 fp16 parts (22 significand bits), 3 partial products on fp16 MFMA, fp32 "
                      "accumulate (the m*m' term is dropped); error vs float64 at the level of torch's own fp32 conv, rms AND "
                      "max (tests/test_kernels_gpu.py::test_conv_fp16_pairs)",
        "config": {"workload": f"RCF stage-1 ResNet50+FCN train step, {B} pairs/GPU {H}x{W} RGB+flow, fp32, SyncBN, Adam (configs[1])",
                   "mask_size": list(mask),
                   "pairs_per_gpu": B, "global_pairs": B * world, "parallelism": f"dp{world}"},
        "loss": round(loss_val, 6),
        "step_tflops_per_gpu": round(value / world * GF_PER_FRAME / 1e3, 2),
        "frac_of_fp32_mfma_roofline": round(value / world * GF_PER_FRAME / 1e3 / FP32_MFMA_PEAK_TF, 4),
        "frac_of_fp16_pair_mfma_roofline": round(value / world * GF_PER_FRAME / 1e3 / H2_MFMA_PEAK_TF, 4),
    }
    # ---- BASELINE configs[2]: the same step with bf16 activations / operands, fp32 master weights and gradients
    bf = None
    if not a.no_bf16:
        fam16 = os.environ.get("RCF_BENCH_FAMILY_BF16", "conv_bf16_dgrad_wide")     # the largest share of the bf16 step (10.6 of 50.6 ms)
        dt16, loss16, prof16, by16, by16s, comm16 = step_leg("bf16", fam16, a.steps, a.warmup)
        v16 = frames / dt16
        bf = {"workload": f"the same step in mixed precision (BASELINE configs[2]): bf16 activations and MFMA operands, fp32 "
                          f"accumulation, fp32 master weights / gradients / Adam; {B} pairs/GPU, dp{world}",
              "dtype": "bf16", "frames_per_s": round(v16, 3), "ms_per_step": round(dt16 / a.steps * 1e3, 3),
              "vs_fp32_step": round(v16 / value, 3), "loss": round(loss16, 6),
              "step_tflops_per_gpu": round(v16 / world * GF_PER_FRAME / 1e3, 2),
              "frac_of_bf16_mfma_roofline": round(v16 / world * GF_PER_FRAME / 1e3 / BF16_MFMA_PEAK_TF, 4)}
        if comm16:
            bf["comm"] = comm16
    # ---- the same step with IEEE fp16 storage (Lightning `precision: 16` of the STv2 / FBMS configs, configs/rcf_stv2/rcf_stage1.yaml:57-60):
    # librcf_hip_f16.so, loss scaling; same kernels as the bf16 step with v_mfma_f32_32x32x16_f16
    f16 = None
    if not a.no_bf16 and not a.no_fp16:
        dtf, lossf, _, _, _, _ = step_leg("fp16", fam16, a.steps, a.warmup)
        vf = frames / dtf
        f16 = {"workload": "the same step with IEEE fp16 activation storage / MFMA operands (Lightning precision: 16), loss scaling, "
                           f"fp32 master weights / gradients / Adam; {B} pairs/GPU, dp{world}", "dtype": "fp16",
               "frames_per_s": round(vf, 3), "ms_per_step": round(dtf / a.steps * 1e3, 3), "loss": round(lossf, 6),
               **getattr(step_leg, "fp16_info", {})}
        out["fp16_step"] = f16
        out["fp16_frames_per_s"], out["fp16_ms_per_step"] = f16["frames_per_s"], f16["ms_per_step"]
    if comm32:
        out["comm"] = comm32
    if rank == 0:
        traffic = None          # HBM bytes per launch from the committed rocprofv3 PMC passes (same workload)
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and (B, H, W) == (8, 480, 854):
            tj = json.load(open(tpath))
            if fam32 in tj.get("by_family", {}):
                traffic = round(tj["by_family"][fam32]["hbm_bytes_per_launch"])
            elif tj.get("family") == fam32:
                traffic = round(tj["hbm_bytes_per_launch"])
        # The convs run as fp32 contractions on the fp16 matrix cores (each operand scaled and split into 2 fp16
        # parts, 3 partial products, fp32 accumulate): the bound is the dense fp16 MFMA peak / 3 passes.
        from rcf_amd import layers as _layers
        two_streams = bool(_layers.SCHED.overlap_wgrad)
        out["config"]["second_stream_for_weight_gradients"] = two_streams
        out["config"]["late_weight_gradients"] = bool(_layers.SCHED.overlap_wgrad and _layers.SCHED.late_wgrad)
        out["config"]["fp16_pair_planes"] = bool(_layers.SCHED.planes)
        out["config"]["bf16_conv_bn_fold"] = bool(_layers.SCHED.fold_bn)        # bf16 leg: conv3 -> bn3 -> join as one tile
        out["schedule"] = _layers.SCHED.as_dict()
        # The timed region runs the step as shipped: weight gradients on a second stream, started after their layer's data
        # gradient.  A bracket there also times what the kernel loses to its neighbour on the chip, so the headline `roofline`
        # is the SAME family bracketed in the separate pass of two steps that runs the whole step on one stream (every launch of
        # the family between HIP events on the stream it is launched on); the timed region's own bracket is kept as `live_*`.
        live = roofline_of(prof, fam32, FAMILIES_F32.get(fam32, fam32), H2_MFMA_PEAK_TF, traffic=traffic)
        src = by32s.get(fam32) if (two_streams and fam32 in by32s and by32s[fam32]["ms"] > 0) else None
        out["roofline"] = roofline_of(src, fam32, FAMILIES_F32.get(fam32, fam32), H2_MFMA_PEAK_TF, traffic=traffic) if src else dict(live)
        out["roofline"].update({"executed_fp16_mfma_tflops": round(3 * out["roofline"]["achieved"], 1),
                                "fp16_mfma_peak": BF16_MFMA_PEAK_TF, "fp32_mfma_peak": FP32_MFMA_PEAK_TF,
                                "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this workload), not measured in this run",
                                "measured": ("one-stream pass after the timed region (2 steps, HIP events around every launch of the family)"
                                             if src else "timed region (one stream)"),
                                "live_achieved": live["achieved"], "live_frac": live["frac"], "live_avg_launch_ms": live["avg_launch_ms"],
                                "live_launches": live["launches"]})
        # matrix-pipe busy fraction and effective clock of the same kernel family: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES /
        # GRBM_GUI_ACTIVE passes over THIS command (tools/pmc_mfma.sh -> profiles/pmc_mfma.json, committed like the traffic):
        # frac ~ mfma_busy x clock_ghz / 2.4 -- the chip clocks down under the convs' power draw (profiles/r06_pmc_headline_random_vs_zero.txt)
        mpath = os.path.join(ROOT, "profiles", "pmc_mfma.json")
        if os.path.exists(mpath) and (B, H, W) == (8, 480, 854):
            mj = json.load(open(mpath)).get("by_family", {})
            if fam32 in mj:
                out["roofline"].update({"mfma_busy": mj[fam32]["mfma_busy"], "clock_ghz": mj[fam32]["clock_ghz"], "l2_hit": mj[fam32]["l2_hit"],
                                        "mfma_busy_source": "profiles/pmc_mfma.json (committed rocprofv3 --pmc passes of this command), not measured in this run"})
            out["pmc_by_family"] = {k: {q: v[q] for q in ("mfma_busy", "clock_ghz", "l2_hit", "launches")} for k, v in mj.items()}
        out["roofline"]["note"] = ("the family with the largest share of the step's kernel time; the deep convs are power-limited "
                                   "(the same launches run 18-23 % faster, at 2.3-2.4 GHz, on all-zero operands: profiles/r06_pmc_headline_random_vs_zero.txt), the family average "
                                   "includes the bottlenecks' short-K 1x1 layers (per-layer table: profiles/)")
        if fam32 in by32s and by32s[fam32]["ms"] > 0:
            out["roofline"]["achieved_one_stream"] = round(by32s[fam32]["flops"] / (by32s[fam32]["ms"] * 1e-3) / 1e12, 2)
            out["roofline"]["frac_one_stream"] = round(out["roofline"]["achieved_one_stream"] / H2_MFMA_PEAK_TF, 4)
        out["roofline_by_kernel"] = by_kernel(by32, FAMILIES_F32, H2_MFMA_PEAK_TF)
        out["roofline_by_kernel_one_stream"] = by_kernel(by32s, FAMILIES_F32, H2_MFMA_PEAK_TF)
        if bf is not None:
            live16 = roofline_of(prof16, fam16, FAMILIES_BF16.get(fam16, fam16), BF16_MFMA_PEAK_TF)
            src16 = by16s.get(fam16) if (two_streams and fam16 in by16s and by16s[fam16]["ms"] > 0) else None
            bf["roofline"] = roofline_of(src16, fam16, FAMILIES_BF16.get(fam16, fam16), BF16_MFMA_PEAK_TF) if src16 else dict(live16)
            bf["roofline"].update({"measured": out["roofline"]["measured"], "live_achieved": live16["achieved"], "live_frac": live16["frac"]})
            if fam16 in by16s and by16s[fam16]["ms"] > 0:
                bf["roofline"]["achieved_one_stream"] = round(by16s[fam16]["flops"] / (by16s[fam16]["ms"] * 1e-3) / 1e12, 2)
                bf["roofline"]["frac_one_stream"] = round(bf["roofline"]["achieved_one_stream"] / BF16_MFMA_PEAK_TF, 4)
            bf["roofline_by_kernel"] = by_kernel(by16, FAMILIES_BF16, BF16_MFMA_PEAK_TF)
            bf["roofline_by_kernel_one_stream"] = by_kernel(by16s, FAMILIES_BF16, BF16_MFMA_PEAK_TF)
            out["bf16_step"] = bf
            # the mixed-precision headline where a reader of the contract keys finds it: top level and inside `roofline`
            out["bf16_frames_per_s"] = bf["frames_per_s"]
            out["bf16_ms_per_step"] = bf["ms_per_step"]
            out["bf16_frac_of_bf16_mfma_roofline"] = bf["frac_of_bf16_mfma_roofline"]
            out["roofline"].update({"bf16_step_frames_per_s": bf["frames_per_s"], "bf16_step_ms": bf["ms_per_step"],
                                    "bf16_step_frac_of_2500TF": bf["frac_of_bf16_mfma_roofline"],
                                    "bf16_kernel_frac": bf["roofline"]["frac"],
                                    "bf16_kernel_frac_one_stream": bf["roofline"].get("frac_one_stream")})
        # CRF ms/frame (second half of BASELINE's metric) -- 480x854, T iterations, batch of 8 frames
        try:
            out["crf_ms_per_frame"] = crf_bench(torch, rcf_amd, synth, dev, H, W, a.crf_iters)
            if world == 1:                                      # the reference's default iteration count (crf_head.py:16)
                out["crf_ms_per_frame_T50"] = crf_bench(torch, rcf_amd, synth, dev, H, W, 50)
                # worst case for the lattice (SURVEY.md §8d "reported separately"): uniform-noise frames, L/N ~ 5.5
                out["crf_ms_per_frame_noise"] = crf_bench(torch, rcf_amd, synth, dev, H, W, a.crf_iters, noise=True)
        except Exception as e:                                  # noqa: BLE001 -- reported, not hidden
            out["crf_ms_per_frame"] = None
            out["crf_error"] = str(e)[:200]
        # backward-warp + photometric L1 residual: HBM-bound, 64 frames per launch
        try:
            out["warp_roofline"] = warp_bench(torch, rcf_amd, synth, dev, H, W)
        except Exception as e:                                  # noqa: BLE001
            out["warp_roofline"] = None
            out["warp_error"] = str(e)[:200]
        # stage 2.1 (BASELINE configs[3]): the same step with CRF self-labels (16 frames through the HIP mean-field
        # CRF, T=5) and the EMA teacher forward + update in the loop
        if not a.no_stage2 and world == 1:
            try:
                out["stage2_step"] = stage2_bench(torch, rcf_amd, config, synth, dev, H, W, B, batch, a.crf_iters, bf16=not a.no_bf16)
            except Exception as e:                              # noqa: BLE001
                out["stage2_step"] = None
                out["stage2_error"] = str(e)[:200]
        # DINO ViT-S/8 forward + soft NCut at 480x856 (the realisable part of BASELINE configs[4])
        if not a.no_stage2 and world == 1:
            try:
                out["vit_s8_ncut"] = vit_bench(torch, rcf_amd, synth, dev)
            except Exception as e:                              # noqa: BLE001
                out["vit_s8_ncut"] = None
                out["vit_error"] = str(e)[:200]
        # the reference's data transform on the device: decoded u8 batch -> collated, normalised batch
        try:
            out["data_transform"] = datapipe_bench(torch, rcf_amd, synth, dev, H, W, B)
        except Exception as e:                                  # noqa: BLE001
            out["data_transform"] = None
            out["data_transform_error"] = str(e)[:200]
        # does the loader keep up?  host JPEG decode + .npy reads + upload + transform, against the bf16 step
        if world == 1:
            try:
                out["loader"] = loader_bench(torch, rcf_amd, synth, dev, H, W, B, step_frames_per_s=out.get("bf16_frames_per_s"))
            except Exception as e:                              # noqa: BLE001
                out["loader"] = None
                out["loader_error"] = str(e)[:200]
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H, W)
        emit(_finite(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


CONTRACT_LINE_MAX = 2048            # the driver keeps an 8 KB stdout tail; round 4's 20.7 KB line could not be parsed


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def contract_line(out):
    """The driver's line: the contract keys of the full result `out`, short strings, no tables.  <= CONTRACT_LINE_MAX bytes."""
    def pick(d, keys):
        return {k: d[k] for k in keys if d is not None and k in d}
    cfg = out.get("config", {})
    line = pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data"))
    line["config"] = {"workload": _clip(cfg.get("workload", ""), 120), **pick(cfg, ("pairs_per_gpu", "global_pairs", "parallelism"))}
    rl = out.get("roofline")
    if rl:
        line["roofline"] = {"kernel": _clip(str(rl.get("kernel", "")).split(" (")[0], 80),
                            **pick(rl, ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms",
                                        "flops_per_launch", "mfma_busy", "clock_ghz"))}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {**pick(cb, ("value", "unit", "cores", "kind", "s_per_step", "timed_steps")),
                                "sample": _clip(cb.get("sample", ""), 160)}
    line.update(pick(out, ("bf16_frames_per_s", "bf16_ms_per_step", "fp16_frames_per_s", "fp16_ms_per_step")))
    bf = out.get("bf16_step") or {}
    if bf.get("roofline"):
        line["bf16_kernel_frac"] = bf["roofline"].get("frac")
    if "frac_of_bf16_mfma_roofline" in bf:
        line["bf16_step_frac"] = bf["frac_of_bf16_mfma_roofline"]
    for key, short in (("crf_ms_per_frame", "crf"), ("crf_ms_per_frame_noise", "crf_noise")):
        c = out.get(key)
        if isinstance(c, dict):
            line[short + "_ms_per_frame"] = c.get("value")
            line[short + "_frac"] = (c.get("roofline") or {}).get("frac")
    if isinstance(out.get("warp_roofline"), dict):
        line["warp_frac"] = out["warp_roofline"].get("frac")
    if isinstance(out.get("stage2_step"), dict):
        line["stage2_ms_per_step"] = out["stage2_step"].get("ms_per_step")
        if out["stage2_step"].get("bf16_ms_per_step") is not None:
            line["stage2_bf16_ms_per_step"] = out["stage2_step"]["bf16_ms_per_step"]
    cm = out.get("comm")
    if cm:                                                      # N > 1: what to read a scaling result against
        line.update(pick(cm, ("syncbn_ms_per_step", "syncbn_collectives", "allreduce_exposed_ms")))
        line["grad_comm"] = _clip(cm.get("grad_communicator", ""), 60)
        cm16 = (out.get("bf16_step") or {}).get("comm")
        if cm16:
            line["bf16_syncbn_ms_per_step"] = cm16.get("syncbn_ms_per_step")
            line["bf16_allreduce_exposed_ms"] = cm16.get("allreduce_exposed_ms")
    line["detail"] = "gpurun_out/bench_detail.json"
    return line


# keys a contract line can lose, in this order, when it would not fit (the contract's own keys are never dropped)
OPTIONAL_KEYS = ("detail", "fp16_ms_per_step", "fp16_frames_per_s", "bf16_allreduce_exposed_ms", "bf16_syncbn_ms_per_step", "grad_comm", "stage2_bf16_ms_per_step",
                 "stage2_ms_per_step", "warp_frac", "crf_noise_frac", "crf_noise_ms_per_frame", "bf16_kernel_frac",
                 "bf16_step_frac", "crf_frac", "crf_ms_per_frame", "bf16_ms_per_step", "bf16_frames_per_s",
                 "allreduce_exposed_ms", "syncbn_collectives", "syncbn_ms_per_step")


def fit_contract_line(line, limit=None):
    """-> the JSON text of `line`, at most `limit` BYTES: optional keys go first, then the strings are clipped harder.  Never
    raises: a benchmark that ran must print its line."""
    limit = limit or CONTRACT_LINE_MAX
    line = dict(line)
    text = json.dumps(line)
    for k in OPTIONAL_KEYS:
        if len(text.encode()) <= limit:
            return text
        line.pop(k, None)
        text = json.dumps(line)
    for n in (80, 40, 16):
        if len(text.encode()) <= limit:
            return text
        for sub, key in (("config", "workload"), ("roofline", "kernel"), ("cpu_baseline", "sample")):
            if isinstance(line.get(sub), dict) and key in line[sub]:
                line[sub][key] = _clip(line[sub][key], n)
        line["metric"] = _clip(line.get("metric", ""), max(n, 48))
        text = json.dumps(line)
    return text


def emit(out):
    """full result -> gpurun_out/bench_detail.json + an earlier, prefixed stdout line; the contract line LAST."""
    full = json.dumps(out)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
            f.write(full + "\n")
    except OSError as e:                                        # a read-only tree must not cost the run its line
        print(f"bench_detail.json not written: {e}", file=sys.stderr)
    print("BENCH_DETAIL " + full, flush=True)
    print(fit_contract_line(contract_line(out)), flush=True)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _self_launch(n):
    """one rank per GPU through torch.distributed.run as a CHILD process (never exec: see the environment notes on
    replacing a process); stdout / stderr are inherited, so rank 0's JSON line is this command's JSON line"""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# bracket families (rcf_amd/ops.py) -> the kernel instance behind each
FAMILIES_F32 = {
    "conv_h2d_fwd": "conv_h2d_kernel<4,false,false> (256-wide column tile; forward of the bottlenecks' convs whose input the batch norm wrote as fp16 pair planes: both operands by LDS-DMA, no split, three LDS stages; incl. the kernels that sum its fused BN statistics)",
    "conv_h2d_fwd_narrow": "conv_h2d_kernel<{1,2},false,false> (the same kernel's 64- / 128-wide column tiles: <= 128 output channels)",
    "conv_h2d_dgrad": "conv_h2d_kernel<4,false,true> (data gradient with dy as fp16 pair planes from the batch norm's backward: the forward kernel on the transposed weights, 256-wide column tile, stride 1)",
    "conv_h2d_dgrad_narrow": "conv_h2d_kernel<{1,2},*,true> / <4,true,true> (the same kernel's 64- / 128-wide column tiles and its strided form)",
    "conv_wgrad_h2d_narrow": "igemm_wgrad_h2d_kernel<2,*,*> (the same weight-gradient kernel on fewer than 256 (tap, channel) columns: 128x128 tile)",
    "conv_wgrad_h2d": "igemm_wgrad_h2d_kernel<4,*,*> (weight gradient with x AND dy as fp16 pair planes: both operands by LDS-DMA, transposing LDS reads, 128x256 tile; incl. the split-K reduction)",
    "conv_h2p_fwd": "conv_h2p_kernel<false> (forward of the deep 3x3 layers: persistent, one wave per SIMD, 4-stage LDS ring, weights by LDS-DMA, fp16 pairs; incl. the kernels that sum its fused BN statistics)",
    "conv_h2p_dgrad": "conv_h2p_kernel<true> (data gradient of the deep 3x3 layers, same kernel)",
    "conv_x3_128x256": "igemm_conv_x3_kernel<2,4,2,2,false,false,2,true,true> (forward, fp16 pairs, 128x256 tile; incl. the two kernels that sum its fused BN statistics)",
    "conv_fwd_narrow": "igemm_conv_x3_kernel<2,{1,2},2,2,...> / fp32-MFMA stem (forward, <= 128 output channels)",
    "conv_dgrad_wide": "igemm_conv_x3_kernel<2,4,2,2,false,true,2,true,true> (data gradient, fp16 pairs, 128x256 tile; incl. the weight transpose+split pre-pass)",
    "conv_dgrad_other": "igemm_conv_x3_kernel<...,true,...> (data gradient: strided / narrow tiles)",
    "conv_wgrad_h2t4": "igemm_wgrad_h2t_kernel<4,*,*,MR> (weight gradient, fp16 pairs, 256x256 (MR = 4: Cout and Cin multiples of 256) or 128x256 tile over (tap, channel) columns, transposing LDS reads, whole tensors and regions; incl. the split-K reduction)",
    "conv_wgrad_other": "igemm_wgrad_h2t_kernel<2,*,*> / igemm_wgrad_x3_kernel / igemm_wgrad_kernel (weight gradient: fewer than 256 columns, odd channel counts, stem)",
}
FAMILIES_BF16 = {
    "conv_bf16_fwd": "conv_bf16_kernel<2,4,2,2,false,false,true,true,3,1,{0,1}> (forward, bf16 operands, 128x256 tile, LDS-DMA loads interleaved with the MFMAs; incl. the fused BN statistics sums; EP 1 = the folded conv3 -> bn3 -> + identity -> ReLU tiles and the x(-T) half of their data gradient)",
    "conv_bf16_fwd_narrow": "conv_bf16_kernel<2,{1,2},2,2,...> (forward, <= 128 output channels)",
    "conv_bf16_dgrad_wide": "conv_bf16_kernel<2,4,2,2,false,true,true,true,3,1,{0,2}> (data gradient, 128x256 tile, LDS-DMA loads; EP 2 = with the previous join's ReLU mask + column sums in the epilogue)",
    "conv_bf16_dgrad_other": "conv_bf16_kernel<...> (data gradient: strided / narrow tiles)",
    "conv_bf16_wgrad4": "wgrad_bf16_dma_kernel<4,false,*> (weight gradient, bf16 operands, 128x256 tile over (tap, channel) columns, LDS-DMA loads, transposing LDS reads; incl. the split-K reduction; also the folded norms' Gram matrices x^T x and G = g^T x)",
    "conv_bf16_wgrad_other": "wgrad_bf16_kernel<1,*,*> / <2,{1,2},*> / <2,4,true> (weight gradient: narrow tiles, regions)",
    # the fp32 stem and the flow head's two small convs keep the fp32 kernels in the bf16 step
    "conv_x3_128x256": FAMILIES_F32["conv_x3_128x256"], "conv_fwd_narrow": FAMILIES_F32["conv_fwd_narrow"],
    "conv_dgrad_other": FAMILIES_F32["conv_dgrad_other"], "conv_wgrad_other": FAMILIES_F32["conv_wgrad_other"],
}


def crf_bench(torch, rcf_amd, synth, dev, H, W, iters, nframes=8, noise=False):
    """CRFHead on `nframes` 480x854 frames per call.  Roofline: algorithmic bytes per frame =
    T*(192 N + 348 L) + build (68 N + 70 L), N pixels, L lattice vertices (measured, data dependent)."""
    import numpy as np
    from rcf_amd.crf import crf_soft_batched
    head = rcf_amd.CRFHead(None, refine_iters=iters)
    make = synth.noise_rgb if noise else synth.smooth_rgb
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4000 + i)) for i in range(nframes)])).to(dev)
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(nframes)])).to(dev)
    for _ in range(3):                                  # untimed: the first call builds with the packed table, from the second on the
        head(imgs, masks)                               # head may pick the sort build (noise-like frames) and loads its kernels once
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        head(imgs, masks)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3 / nframes
    rgb, unary = head.prepare(imgs, masks)
    _, nv = crf_soft_batched(rgb, unary, W, H, head.scomp_smooth, head.sxy_smooth, head.scomp, head.sxy, head.srgb, 1,
                             want_nvert=True)
    N, L = H * W, float(nv[:, 1].float().mean())
    alg = iters * (192.0 * N + 348.0 * L) + 68.0 * N + 70.0 * L
    ach = alg / (ms * 1e-3) / 1e9
    return {"iters": iters, "frames_per_call": nframes, "frames": "uniform noise (worst case)" if noise else "smooth synthetic",
            "lattice_build": "sort" if head.last_build == 3 else "packed hash table",
            "value": round(ms, 4), "pixels": N, "lattice_vertices": round(L),
            "algorithmic_bytes_per_frame": round(alg),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4)}}


def stage2_bench(torch, rcf_amd, config, synth, dev, H, W, B, batch, iters, steps=3, bf16=True):
    """the stage-2.1 step (the one BASELINE configs[3] trains with the CRF in the loop, models/rcf_model.py:490-529) in fp32 and,
    `bf16`, in mixed precision (parity: tests/test_stage2_gpu.py::test_stage21_bf16_step_vs_reference_autocast)"""
    import types
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=1, eval_save=False, eval_export=False)

    def leg(precision, iters=iters):
        model = rcf_amd.RCFModel(args, **config.stage21_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="BN",
                                                                     refine_iters=iters))
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
        tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=dev, precision=precision)
        for _ in range(PRIMING_STEPS):                 # allocator / second-stream scratch reach their steady state
            tr.step(batch)
        torch.cuda.synchronize()
        times = []
        for _ in range(steps):                         # secondary leg: median of individually timed steps
            t0 = time.perf_counter()
            losses = tr.step(batch)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        lc = float(losses["loss_crf"])
        del tr, model
        torch.cuda.empty_cache()
        return sorted(times)[len(times) // 2], lc
    dt, lc = leg(None)
    out = {"workload": f"stage 2.1 step: stage-1 step + EMA teacher forward + CRF (T={iters}) on {2 * B} frames + EMA update",
           "ms_per_step": round(dt * 1e3, 2), "frames_per_s": round(2 * B / dt, 2), "loss_crf": round(lc, 6)}
    if bf16:
        dt16, lc16 = leg("bf16")
        out.update({"bf16_ms_per_step": round(dt16 * 1e3, 2), "bf16_frames_per_s": round(2 * B / dt16, 2),
                    "bf16_loss_crf": round(lc16, 6)})
        # the reference's own stage-2.1 config leaves CRFHead at its default of T = 50 iterations (configs/rcf/rcf_stage2.1.yaml:150-151,
        # models/crf_head.py:13): the CRF runs on the second stream beside the student's forward
        dt50, _ = leg("bf16", 50)
        out["bf16_ms_per_step_crf_T50"] = round(dt50 * 1e3, 2)
    return out


def vit_bench(torch, rcf_amd, synth, dev, frames=4):
    from rcf_amd import ncut, vit
    m = vit.vit_small(patch_size=8)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_vit_state_dict(shapes, seed=21).items()})
    m = m.to(dev).eval()
    x = torch.randn(frames, 3, 480, 856, device=dev)
    T = 60 * 107 + 1
    gf = (12 * (2 * T * (384 * 1152 + 384 * 384 + 2 * 384 * 1536) + 4 * T * T * 384) + 2 * 6420 * 384 * 192) / 1e9

    def timeit(fn, n=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    t_fwd = timeit(lambda: m(x)) / frames
    feats = m.get_last_qkv(x[:1], "k")
    mask = (torch.rand(60, 107, device=dev) > 0.5).float() * 0.8 + 0.1
    t_nc = timeit(lambda: ncut.ncut_refine(feats, mask, steps=10, learning_rate=0.45))
    return {"workload": f"DINO ViT-S/8 forward at 480x856 ({frames} frames per call, 6421 tokens, 12 blocks, fused attention and "
                        "linear layers on fp16 pairs: fp32-level error) + soft NCut (6420^2 affinity, 10 Adam steps)", "vit_ms_per_frame": round(t_fwd * 1e3, 2),
            "vit_gflop_per_frame": round(gf, 1), "vit_tflops": round(gf / 1e3 / t_fwd, 1),
            "ncut_refine_ms_per_frame": round(t_nc * 1e3, 2)}


def warp_bench(torch, rcf_amd, synth, dev, H, W, nframes=64):
    """flow_warp fused with the occlusion-masked L1 residual on `nframes` frames per launch: 4*(2 + 2*3 + 1)
    algorithmic bytes per pixel (flow, source taps once, target, occlusion mask; scalar output)."""
    import numpy as np
    from rcf_amd import ops
    base = np.stack([synth.voronoi_affine_flow(H, W, 7000 + i)[0] for i in range(8)])
    fl = torch.from_numpy(np.tile(base, (nframes // 8, 1, 1, 1))).to(dev)
    x = torch.rand(nframes, 3, H, W, device=dev)
    y = torch.rand(nframes, 3, H, W, device=dev)
    occ = torch.ones(nframes, 1, H, W, device=dev)

    def timeit(fn, n=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    px = nframes * H * W
    t_l1 = timeit(lambda: ops.warp_l1_residual(y, x, fl, occ, "border"))
    t_w = timeit(lambda: ops.flow_warp(x, fl, "border"))
    ach = px * 36.0 / t_l1 / 1e9
    return {"kernel": "warp_l1_kernel (bilinear backward warp + masked L1 residual)", "frames_per_launch": nframes,
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "us_per_frame": round(t_l1 / nframes * 1e6, 2),
            "flow_warp_GBps": round(px * 32.0 / t_w / 1e9, 1), "flow_warp_us_per_frame": round(t_w / nframes * 1e6, 2)}


def _finite(o):
    """strict JSON: a non-finite float (a diverged loss) becomes null instead of NaN / Infinity"""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def datapipe_bench(torch, rcf_amd, synth, dev, H, W, B):
    """rcf_amd.data_pipeline.Transform(training, strong_aug, has_pl) on a batch of B decoded samples (2 frames, 2 flows,
    2 pseudo-label masks of HxW each) resident in HBM -> 2 x [B,3,384,384] + flows + masks.  Algorithmic bytes: what is
    written plus the source pixels the crops cover once (frames 3 B, flows 8 B, masks 1 B per source pixel)."""
    import numpy as np
    from rcf_amd.data_pipeline import Transform
    base = [synth.loader_sample(9000 + i, H, W) for i in range(2)]
    samples = [base[i % 2] for i in range(B)]
    tf = Transform(training=True, strong_aug=True, has_pl=True)
    rng = np.random.RandomState(5)
    params = np.stack([tf.sample_params(H, W, rng) for _ in range(B)])
    params["ops"] |= 15                                       # every photometric stage on: the most arithmetic per pixel
    data = {"imgs": torch.from_numpy(np.stack([s["frames"] for s in samples])).to(dev),
            "gt_fw_flows": torch.from_numpy(np.stack([s["fw"] for s in samples])).to(dev),
            "gt_bw_flows": torch.from_numpy(np.stack([s["bw"] for s in samples])).to(dev),
            "pl_masks": torch.from_numpy(np.stack([s["pl"] for s in samples])).to(dev)}
    tf(data, params=params)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        tf(data, params=params)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    out_b = B * 384 * 384 * (2 * 3 * 4 + 2 * 2 * 4 + 2 * 4)
    src_b = sum(384.0 * 384.0 * (H / float(p["rh"])) * (W / float(p["rw"])) * (2 * 3 + 2 * 8 + 2 * 1) for p in params)
    ach = (out_b + src_b) / (ms * 1e-3) / 1e9
    return {"workload": f"Transform(training, strong_aug, has_pl) on {B} samples of 2 frames {H}x{W} (+2 flows, 2 masks), "
                        f"every photometric stage on", "ms_per_batch": round(ms, 4), "samples_per_s": round(B / (ms * 1e-3), 1),
            "launches_per_batch": 3, "algorithmic_bytes_per_batch": round(out_b + src_b),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4)}}


def loader_bench(torch, rcf_amd, synth, dev, H, W, B, workers=16, batches=10, step_frames_per_s=None):
    """SURVEY.md section 8(f) rank 4's question: does the loader keep up with the step?  The path the reference's DataLoader
    walks (dataset/data.py:62-70,122-128; 16 workers: configs/rcf/rcf_stage1.yaml:10) with this repo's device side: JPEG files
    -> PIL decode (host, `workers` threads: the decoder and the file reads release the GIL) and `.npy` flows read straight
    into the pinned staging set of `BatchUploader` -> copy stream -> `Transform` (three launches).  Files are written to a
    temporary directory first and stay in the page cache (what a training run sees from its second epoch on); the GPU does
    nothing else meanwhile.  frames/s = 2 B per batch over the wall time of `batches` batches after 2 untimed ones."""
    import concurrent.futures as cf
    import os
    import shutil
    import tempfile
    import time
    import numpy as np
    from PIL import Image
    from rcf_amd.data_pipeline import BatchUploader, Transform, load_flow_npy_into
    tmp = tempfile.mkdtemp(prefix="rcf_loader_")
    try:
        nsamp = 2 * B
        jpeg_bytes = 0
        for i in range(nsamp):
            smp = synth.loader_sample(7000 + i, H, W)
            for f in range(2):
                fp = os.path.join(tmp, f"{i}_{f}.jpg")
                Image.fromarray(smp["frames"][f]).save(fp, quality=95)
                jpeg_bytes += os.path.getsize(fp)
            np.save(os.path.join(tmp, f"{i}_fw.npy"), smp["fw"])
            np.save(os.path.join(tmp, f"{i}_bw.npy"), smp["bw"])
        tf = Transform(training=True, strong_aug=True, has_flow=True)
        up = BatchUploader(B, 2, H, W, has_flow=True, has_pl=False, device=dev)
        rng = np.random.RandomState(3)

        def fill(job):                                          # one FILE per task: 4 B tasks per batch keep 16 workers busy
            st, b, i, what = job
            if what in (0, 1):
                with Image.open(os.path.join(tmp, f"{i}_{what}.jpg")) as im:
                    st["imgs"][b, what] = np.asarray(im.convert("RGB"))
            else:
                load_flow_npy_into(os.path.join(tmp, f"{i}_{what}.npy"), st["gt_" + what + "_flows"][b])

        def one(it, pool):
            st = up.stage()
            list(pool.map(fill, [(st, b, (it * B + b) % nsamp, what) for b in range(B) for what in (0, 1, "fw", "bw")]))
            params = np.stack([tf.sample_params(H, W, rng) for _ in range(B)])
            return tf(up.upload(), params=params)
        out = {}
        with cf.ThreadPoolExecutor(workers) as pool:
            for it in range(2):
                one(it, pool)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            host = 0.0
            for it in range(batches):
                h0 = time.perf_counter()
                one(2 + it, pool)
                host += time.perf_counter() - h0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        fps = 2 * B * batches / dt
        out = {"what": f"{H}x{W} JPEG pairs (q95, {jpeg_bytes / (2 * nsamp) / 1e3:.0f} KB per frame, page cache) + 2 .npy flows per sample: "
                       f"PIL decode on {workers} host threads -> pinned staging -> copy stream -> Transform(training, strong_aug) on the GPU; "
                       f"one batch of {B} samples in flight at a time",
               "workers": workers, "host_cores": os.cpu_count(), "frames_per_s": round(fps, 1), "ms_per_batch": round(1e3 * dt / batches, 2),
               "host_ms_per_batch": round(1e3 * host / batches, 2), "uploaded_bytes_per_batch": up.nbytes}
        if step_frames_per_s:
            out["vs_bf16_step_frames_per_s"] = round(fps / step_frames_per_s, 3)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _cpu_info():
    """(model name, logical CPUs available to this process, physical cores among them)"""
    model, phys = "unknown", set()
    try:
        allowed = os.sched_getaffinity(0)
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur and int(cur.get("processor", -1)) in allowed:
                    phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
                continue
            k, v = [x.strip() for x in line.split(":", 1)]
            cur[k] = v
            if k == "model name":
                model = v
        if cur and int(cur.get("processor", -1)) in allowed:
            phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
        return model, len(allowed), max(len(phys), 1)
    except Exception:                                           # noqa: BLE001
        return model, os.cpu_count() or 1, os.cpu_count() or 1


def cpu_baseline(H, W):
    """BASELINE.md section 4 / configs[0]: the oracle (CPU restatement of the reference, pinned to it in the build
    container) on the GPU box's host cores, as that section prescribes -- 4 pairs of 480x854, fp32, ALL physical cores, one
    untimed warm-up step + 3 timed training steps (fwd + bwd + Adam), s/step = their mean -- plus the sequential C
    restatement of tools/torchCRF (oracle/crf_ref.c) on one 480x854 frame at T = 5 and T = 50.  Bounded: when the warm-up
    step alone takes more than 60 s (a small host) only one step is timed, and hosts short of memory take 2 or 1 pairs; the
    `sample` field says what ran."""
    import copy
    import types
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import crf_oracle
    import rcf_torch as orc
    from rcf_amd import config, synth
    model_name, logical, physical = _cpu_info()
    cores = physical
    torch.set_num_threads(cores)
    avail_gb = 0.0
    try:
        avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
    except Exception:                                           # noqa: BLE001
        pass
    pairs = 4 if avail_gb >= 48 else (2 if avail_gb >= 28 else 1)       # the B=4 step peaks at ~20 GB resident
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None)
    m = orc.RCFModel(args, **copy.deepcopy(config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="BN")))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    m.train()
    opt = orc.make_optimizer(m, 1e-4, 1e-4)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x))

    def step(n):
        nb = synth.make_batch(n, H, W, config_id=2)
        batch = {"imgs": [t(x) for x in nb["imgs"]], "gt_fw_flows": [t(x) for x in nb["gt_fw_flows"]],
                 "gt_bw_flows": [t(x) for x in nb["gt_bw_flows"]]}
        t0 = time.perf_counter()
        losses = m(batch)
        opt.zero_grad()
        losses["loss"].backward()
        opt.step()
        return time.perf_counter() - t0
    warm = step(pairs)
    nsteps = 3 if warm <= 60.0 else 1
    times = [step(pairs) for _ in range(nsteps)]
    dt = sum(times) / len(times)
    # CPU CRF: the restatement of tools/torchCRF is sequential C (1 core)
    rgb, msk = synth.smooth_rgb(H, W, 4000), synth.soft_blob_mask(H, W, 4000)
    img = torch.from_numpy(synth.normalize_rgb(rgb))[None]
    crf = {}
    for iters in (5, 50):
        head = orc.CRFHead(None, refine_iters=iters, crf_soft=crf_oracle.crf_soft_torch)
        t0 = time.perf_counter()
        head(img, torch.from_numpy(msk)[None])
        crf[f"crf_ms_per_frame_T{iters}"] = round((time.perf_counter() - t0) * 1e3, 1)
    # CPU data transform: the numpy restatement of the reference's per-sample pipeline (oracle/transforms_np.py), 1 core
    from oracle import transforms_np as T_np
    from rcf_amd.data_pipeline import Transform
    tfm = Transform(training=True, strong_aug=True, has_pl=True)
    smp = synth.loader_sample(9000, H, W)
    prm = tfm.sample_params(H, W, np.random.RandomState(5))
    prm["ops"] |= 15
    t0 = time.perf_counter()
    T_np.apply_params(smp, prm, 384, 384)
    crf["data_transform_ms_per_sample"] = round((time.perf_counter() - t0) * 1e3, 1)
    return {"value": round(2.0 * pairs / dt, 4), "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "cpu_model": model_name, "logical_cpus": logical, "physical_cores": physical,
            "s_per_step": round(dt, 2), "timed_steps": nsteps, "steps_s": [round(x, 2) for x in times],
            "sample": f"{pairs} pairs {H}x{W} (configs[0]), {nsteps} timed fwd+bwd+Adam steps of oracle/rcf_torch.py after 1 warm-up "
                      f"({warm:.0f} s), {cores} threads = all physical cores",
            **crf, "crf_cores": 1, "crf_kind": "port (oracle/crf_ref.c, sequential C restatement of tools/torchCRF)",
            "data_transform_kind": "port (oracle/transforms_np.py: numpy restatement of dataset/transforms.py's per-sample pipeline, 1 core)"}


if __name__ == "__main__":
    main()
