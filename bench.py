#!/usr/bin/env python
"""bench.py - MiCo omni-modal alignment step (ViT-g/14 fwd+bwd) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 20 --warmup 3          (launches its own 8 ranks, one per GPU, when not already under torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the single-GPU ViT-g/14 configuration the metric is quoted on): per GPU b = 64 samples
of image (1 frame, 224^2) + audio (4 spectrogram windows of 224x224 = 10 s of mel-spec) + text (77 tokens), synthetic
inputs resident in HBM, random-init weights of the real architecture (EVA01-CLIP-g/14 tower shared by image and audio,
BERT-base with cross-attention), one step = forward + backward of the full alignment loss "ret%tva_cap%tva" (ITC + ITM with
in-batch hard negatives + causal masked-caption LM; step-B of SURVEY.md section 8d).  For N > 1 every rank runs that workload on
its own shard (weak scaling), the contrastive features are exchanged with one packed RCCL all-gather, hard-negative condition
rows with an index-then-fetch all-to-all, and gradients are averaged with bucketed all-reduces overlapped with backward.

Precision of the timed run (--dtype): fp16 MFMA operands, fp32 accumulation / residual stream / statistics - the 16-bit type the
reference's own trainer runs in (fp16 autocast, data/utils/pipeline.py:43) at the bf16 MFMA rate.  `parity` in the JSON line holds
max|out - ref| / max|ref| of exactly this configuration against the reference-generated goldens (tests/golden/vit_g14_*.pt), measured
in the same process after the timed region; `parity_config` is the same step timed in the configuration with margin under the 1e-3
gate (fp16 with hi/lo-split weights: 2 k-segments per forward GEMM) next to it.

One JSON line on rank 0 (see the repo prompt for the field contract) with extra objects:
  roofline     - the dominant kernel (MFMA GEMM) timed per launch with HIP events inside the timed region;
  cpu_baseline - the CPU oracle (oracle/mico_oracle.py, a restatement of the reference parity-locked to it) timed on the
                 host cores on a bounded sample of the same workload plus the BASELINE.md section 4 anchors (rank 0, N = 1 only);
  parity / parity_config / secondary / comm - see above and main().
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic forward GF / sample for step-B at config-3 shapes: 2 * MAC of every
# GEMM incl. attention, LM head only where a loss consumes it; fwd+bwd = 3 x fwd
ALG_TFLOP_PER_SAMPLE = {"vitg_img1_aud4_txt77_stepB": 8.74, "vitg_omni14_txt77": 24.04, "vitg_vid8_cap": 13.08}
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md

PRECISIONS = {   # --dtype -> (torch dtype, split_fp16, split_mode, description)
    "fp8": ("bfloat16", False, "full", "block-scaled MX fp8 (e4m3, E8M0 per 32) MFMA for the forward and input-gradient GEMMs of the towers and "
                                       "BERT's large projections; weight gradients, attention and everything else as bf16"),
    "fp16": ("float16", False, "full", "fp16 MFMA operands, fp32 accumulate / residual stream / LN / softmax; forward GEMMs of the first 4 of the "
                                       "40 tower blocks as x W_hi + x W_lo (runtime.CFG.head_split_blocks)"),
    "fp16-plain": ("float16", False, "full", "fp16 MFMA operands (plain everywhere), fp32 accumulate / residual stream / LN / softmax"),
    "bf16": ("bfloat16", False, "full", "bf16 MFMA operands, fp32 accumulate / residual stream / LN / softmax"),
    "fp16-split-w": ("float16", True, "weights", "fp16 MFMA, forward GEMMs x W_hi + x W_lo (weights hi/lo split, 2 k-segments)"),
    "fp16-split": ("float16", True, "full", "fp16 MFMA, forward GEMMs x_hi W_hi + x_lo W_hi + x_hi W_lo (3 k-segments)"),
}


HEAD_SPLIT_BLOCKS = 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--vision", default="evaclip01_giant")
    ap.add_argument("--layers", type=int, default=None, help="truncate the ViT (debug only; invalidates the metric)")
    ap.add_argument("--dtype", default="fp16", choices=sorted(PRECISIONS))
    ap.add_argument("--task", default=None)
    ap.add_argument("--workload", default="img_aud_txt", choices=["img_aud_txt", "omni", "vid_cap_fp8"],
                    help="img_aud_txt = BASELINE configs[2] (the metric's single-GPU configuration, default); omni = one rank's share "
                         "of configs[3]: image+video (9 vision frames) + depth + audio + text, 14 frames/sample (not the headline metric); "
                         "vid_cap_fp8 = one rank of configs[4]: 8 video frames + BERT generative head (CAP), b = 32, --dtype fp8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--diet", type=int, default=None, help="force the tower's saved-activation level (0 / 1 / 2; default: mico_amd.functional.tower_plan decides)")
    ap.add_argument("--no-comm", action="store_true", help="N = 1: skip the extra steps on a one-rank RCCL group (the `comm` object)")
    ap.add_argument("--no-extras", action="store_true", help="skip parity / parity_config / secondary (only the headline measurement)")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--eval-mode", action="store_true", help="disable DropPath (parity-style run)")
    ap.add_argument("--gemm-detail", action="store_true", help="print a per-shape table of the timed GEMM launches to stderr")
    ap.add_argument("--no-gemm-timer", action="store_true", help="A/B switch: no per-launch HIP events around the GEMMs (roofline is then null)")
    ap.add_argument("--optimizer", action="store_true",
                    help="also run the AdamW step (mico_amd.optim, SURVEY section 8 row f4) inside the timed step: a full training step, "
                         "beyond the metric's fwd+bwd definition")
    ap.add_argument("--no-bert-dropout", action="store_true", help="A/B switch: BERT dropout probabilities set to 0 (invalidates the metric)")
    ap.add_argument("--dense-droppath", action="store_true",
                    help="evaluate dropped residual branches too and multiply them by 0 (the reference's schedule) instead of skipping them")
    return ap.parse_args()


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd_cpu, args):
    """The CPU oracle on the host cores.  `value`: the bench's own step (same task, same shapes) at --cpu-batch samples, one step
    (20-30 s of CPU work).  `anchors`: the BASELINE.md section 4 shapes - config 2 (ViT-B/16 image + text contrastive step) at bs 8,
    config 3's image + text sub-step (ViT-g/14) at bs 2, config 1 (single-image ViT-g/14 encode) - one warm-up + 3 timed steps,
    median.  32 threads: PyTorch's CPU GEMMs on this path stop scaling (and regress badly) far below the 256 hardware threads of the
    GPU box's host - a first run with all 256 threads took 1292 s for the same sample."""
    from oracle import mico_oracle as O
    from mico_amd.weights import synth_inputs, synth_state_dict
    from mico_amd.model import MiCo, default_cfg
    import random
    ncores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    arch = O.ARCHS[args.vision]
    b = args.cpu_batch

    def tied(sd):
        sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        sd["multimodal_encoder.cls.predictions.decoder.weight"] = sd["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
        return sd

    inp = synth_inputs(dict(b=b, **WORKLOADS[args.workload]["shape"]), seed=99)
    sd = tied(sd_cpu)
    mi, lab = O.token_masker(inp["input_ids"], 0.6, random.Random(0))
    idx = torch.arange(b).roll(1)
    injected = {st: dict(neg_cond_idx=idx, neg_text_idx=idx) for st in ("tva", "tvd", "tv")}
    injected["cap"] = dict(masked_ids=mi, labels=lab)
    t0 = time.time()
    out, _ = O.mico_forward(sd, arch, inp, args.task, dict(itm_ratio=0.1), injected=injected)
    sum(out.values()).backward()
    dt = time.time() - t0
    res = dict(value=b / dt, unit="samples/s", cores=ncores, kind="port", cpu=cpu_model_string(), host_threads=os.cpu_count(),
               sample=f"oracle/mico_oracle.py fp32, same step ({args.task}) at b={b} ({WORKLOADS[args.workload]['frames']} frames + 77 tokens "
                      f"per sample), 1 step, {dt:.1f} s on {ncores} threads")

    def timed(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t = time.time()
            fn()
            ts.append(time.time() - t)
        return statistics.median(ts)

    def itc_step(vt, sdx, bs):
        inpx = synth_inputs(dict(b=bs, vision=1, S=77), seed=98)

        def f():
            for v in sdx.values():
                v.grad = None
            o, _ = O.mico_forward(sdx, O.ARCHS[vt], inpx, "ret%tv", dict(itm_ratio=0.1),
                                  injected={"tv": dict(neg_cond_idx=torch.arange(bs).roll(1), neg_text_idx=torch.arange(bs).roll(1))})
            o["loss_itc"].backward()
        return f

    anchors = {}
    mb = MiCo(default_cfg("evaclip02_base"))
    sdb = tied(synth_state_dict({k: tuple(v.shape) for k, v in mb.state_dict().items()}, seed=0))
    del mb
    t = timed(itc_step("evaclip02_base", sdb, 8))
    anchors["config2_vitb16_img_txt_itc_bs8"] = dict(samples_per_s=8 / t, median_step_s=t)
    del sdb
    t = timed(itc_step(args.vision, sd, 2))
    anchors["config3_vitg14_img_txt_itc_bs2"] = dict(samples_per_s=2 / t, median_step_s=t)
    px = synth_inputs(dict(b=1, vision=1, S=0), seed=97)["vision_pixels"]

    def enc():
        with torch.no_grad():
            O.encode_batch(sd, arch, dict(vision_pixels=px))
    t = timed(enc)
    anchors["config1_vitg14_single_image_encode"] = dict(samples_per_s=1 / t, median_step_s=t)
    res["anchors"] = anchors
    res["anchors_protocol"] = "BASELINE.md section 4: fp32, 1 warm-up + 3 timed steps, median"
    return res


WORKLOADS = {
    "img_aud_txt": dict(shape=dict(vision=1, audio=4, S=77), task="ret%tva_cap%tva", key="vitg_img1_aud4_txt77_stepB", frames=5),
    "omni": dict(shape=dict(vision=9, depth=1, audio=4, S=77), task="ret%tva%tvd_cap%tva", key="vitg_omni14_txt77", frames=14),
    "vid_cap_fp8": dict(shape=dict(vision=8, S=77), task="cap%tv", key="vitg_vid8_cap", frames=8, batch=32, dtype="fp8"),
}


def set_precision(name):
    from mico_amd import runtime
    dt, split, mode, _ = PRECISIONS[name]
    runtime.set_compute_dtype(getattr(torch, dt))
    runtime.CFG.split_fp16, runtime.CFG.split_mode = split, mode
    runtime.CFG.fp8 = name == "fp8"
    runtime.CFG.head_split_blocks = HEAD_SPLIT_BLOCKS if name == "fp16" else 0
    runtime.clear_weight_cache()


def measure_parity(model, dev):
    """max|out - ref| / max|ref| of the ACTIVE precision configuration against the reference-generated goldens: the depth-2 g/14 tower
    (final-LN tokens of 2 images; its own 2-block model with the same weight generator) and the full 40-block g/14 (token rows + the
    L2-normalised 512-d feat_v; needs the bench's full-depth model)."""
    from mico_amd.model import MiCo, default_cfg
    from mico_amd.weights import synth_state_dict
    from mico_amd.functional import l2_normalize
    gd = os.path.join(ROOT, "tests", "golden")
    out = {}
    was_training = model.training
    try:
        fx = torch.load(os.path.join(gd, "vit_g14_d2.pt"), map_location="cpu", weights_only=False)
        m2 = MiCo(default_cfg("evaclip01_giant", vision_layers=2))
        m2.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m2.state_dict().items()}, 0), strict=False)
        m2.to(dev).eval()
        g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
        x = torch.randn((2, 3, 224, 224), generator=g)
        with torch.no_grad():
            o = m2.vision_encoder.visual(x.to(dev), return_all_features=True).float().cpu()
        out["vit_g14_depth2_tokens"] = ((o - fx["out"]).abs().max() / fx["out"].abs().max()).item()
        del m2
        if model.config.get("vision_layers") is None and model.config.vision_encoder_type == "evaclip01_giant":
            fx = torch.load(os.path.join(gd, "vit_g14_full.pt"), map_location="cpu", weights_only=False)
            g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
            x = torch.randn((1, 1, 3, 224, 224), generator=g).to(dev)
            model.eval()
            with torch.no_grad():
                o = model.forward_vision_encoder(x)
                feat = l2_normalize(model.contra_head_v(model.pool_vision_for_contra(o))).float().cpu()
            out["vit_g14_full_token_rows"] = ((o[0, 0, [0, 1, 128, 256]].float().cpu() - fx["rows"]).abs().max() / fx["amax"]).item()
            out["vit_g14_full_feat_v"] = ((feat - fx["feat_v"]).abs().max() / fx["feat_v"].abs().max()).item()
    finally:
        model.train(was_training)
    out["metric"] = "max|out - ref| / max|ref| vs reference fp32 CPU outputs (tests/golden, generated by oracle/make_golden.py)"
    out["gate"] = 1e-3
    return out


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become the launcher - one process per GPU over RCCL, same arguments."""
    n = torch.cuda.device_count()
    if n < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {n} GPU(s) are visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    wl = WORKLOADS[args.workload]
    args.task = args.task or wl["task"]
    if "dtype" in wl and "--dtype" not in " ".join(sys.argv):
        args.dtype = wl["dtype"]
    if "batch" in wl and "--batch" not in " ".join(sys.argv):
        args.batch = wl["batch"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes a version banner to the C-level stdout when its first
    # communicator comes up, flushed at exit - i.e. AFTER the JSON line): keep the real stdout aside, point fd 1 at stderr for the whole run
    # and write the line to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without torchrun)")
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from mico_amd import runtime, ops
    from mico_amd.model import MiCo, default_cfg
    from mico_amd.weights import synth_state_dict, synth_inputs
    from mico_amd.distributed import GradBucketReducer, packed_all_gather

    set_precision(args.dtype)
    runtime.set_activation_diet(args.diet)
    torch.manual_seed(rank)     # host RNG: stochastic-depth draws differ per rank (weights/inputs come from counter hashes)
    from mico_amd.functional import DropPlan
    DropPlan.skip_dropped = not args.dense_droppath
    cfg = default_cfg(args.vision, vision_layers=args.layers)
    model = MiCo(cfg)
    sd_cpu = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd_cpu, strict=False)
    if args.no_bert_dropout:
        model.multimodal_encoder.bert.config.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model.to(dev)
    model.eval() if args.eval_mode else model.train()
    b = args.batch
    batch = {k: v.to(dev) for k, v in synth_inputs(dict(b=b, **wl["shape"]), seed=1234 + rank).items()}
    reducer = GradBucketReducer(model.parameters()) if world > 1 else None
    optimizer = None
    if args.optimizer:
        from mico_amd.optim import AdamW
        decay = [p for n, p in model.named_parameters() if not any(k in n for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
        nodecay = [p for n, p in model.named_parameters() if any(k in n for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
        optimizer = AdamW([dict(params=decay, weight_decay=0.01, lr=1e-6), dict(params=nodecay, weight_decay=0.0, lr=1e-6)],
                          lr=1e-6, betas=(0.9, 0.98))
    finish_ms = []

    def step(the_batch=None, task=None):
        model.zero_grad(set_to_none=True)
        losses = model(dict(the_batch if the_batch is not None else batch), task or args.task, compute_loss=True)
        total = sum(losses.values())
        total.backward()
        if reducer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reducer.finish()
            e1.record()
            finish_ms.append((e0, e1))
        if optimizer is not None:
            optimizer.step()
        return losses

    def timed_steps(n, the_batch=None, task=None):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            losses = step(the_batch, task)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el, losses

    for _ in range(args.warmup):
        step()
    timer = None if args.no_gemm_timer else ops.KernelTimer()
    ops.GEMM_TIMER = timer
    DropPlan.stats[:] = [0, 0]
    finish_ms.clear()
    elapsed, losses = timed_steps(args.steps)
    ops.GEMM_TIMER = None
    kept = DropPlan.stats[0] / DropPlan.stats[1] if DropPlan.stats[1] else 1.0
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    if args.gemm_detail and rank == 0 and timer is not None:
        tab = {}
        for (var, flops, e0, e1), det in zip(timer.records, timer.detail):
            d = tab.setdefault((var, det[1], det[2], det[3], det[4]), [0, 0.0, 0.0, 0])
            d[0] += 1; d[1] += flops; d[2] += e0.elapsed_time(e1); d[3] += det[0]
        print(f"{'(ta,tb,kernel)':>14s} {'N':>6s} {'K':>6s} {'epilogue':>10s} {'split':>5s} {'launches':>8s} {'avg M':>8s} {'avg us':>8s} {'TFLOP/s':>8s} {'ms/step':>8s}", file=sys.stderr)
        for key, d in sorted(tab.items(), key=lambda kv: -kv[1][2]):
            print(f"{str(key[0]):>14s} {key[1]:6d} {key[2]:6d} {key[3]:>10s} {key[4]:5d} {d[0]:8d} {d[3] / d[0]:8.0f} {d[2] / d[0] * 1e3:8.1f} "
                  f"{d[1] / d[2] / 1e9:8.1f} {d[2] / args.steps:8.2f}", file=sys.stderr)

    # ---- communication figures: exposed gradient-reduction wait per step, packed all-gather latency.  N > 1: from the timed steps.
    # N = 1: the SAME code on a one-rank RCCL group (mico_amd.distributed.force_dist: packed_all_gather's all_gather_into_tensor,
    # fetch_rows' all_to_all_single, the reducer's in-place all_reduce(AVG) of the tower's arena slices + its buckets), a few extra steps
    # after the headline measurement - what a 1-GPU box can say about the N > 1 path: that it runs on RCCL and what its overhead is.
    comm = None
    forced_ms = None
    if world == 1 and not args.no_comm and rank == 0:
        from mico_amd import distributed as D
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            D.force_dist(True)
            reducer = GradBucketReducer(model.parameters())
            step()
            finish_ms.clear()
            kc = max(2, args.steps // 2)
            elc, _ = timed_steps(kc)
            forced_ms = elc / kc * 1e3
        except Exception as e:
            comm = {"error": repr(e)}
    if (world > 1 or forced_ms is not None) and comm is None:
        exposed = sum(e0.elapsed_time(e1) for e0, e1 in finish_ms) / max(1, len(finish_ms))
        feat = torch.randn(b, 512, device=dev)
        ids = batch["input_ids"]
        for _ in range(3):
            packed_all_gather([feat, ids, batch["attention_mask"], feat])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            packed_all_gather([feat, ids, batch["attention_mask"], feat])
        torch.cuda.synchronize()
        ag_us = (time.perf_counter() - t0) / 20 * 1e6
        grad_bytes = sum(p.numel() for p in model.parameters() if p.requires_grad) * 4
        comm = dict(backend="nccl (RCCL)", world_size=dist.get_world_size(), packed_allgather_us=ag_us,
                    packed_allgather_bytes_per_rank=int(feat.numel() * 4 * 2 + ids.numel() * 8 * 2),
                    grad_bytes=grad_bytes, grad_reduce_exposed_ms_per_step=exposed,
                    grad_reduce_note="time the step spends in GradBucketReducer.finish() waiting for reductions that did not hide behind "
                                     "the backward; the ViT blocks' arena slices are reduced in place from inside the backward")
        if forced_ms is not None:
            comm.update(forced_at_world_size_1=True, ms_per_step_with_collectives=forced_ms, steps=kc,
                        ms_per_step_headline=elapsed / args.steps * 1e3,
                        note="one-rank RCCL group with the N > 1 code paths forced (MICO_FORCE_DIST semantics): every collective of the "
                             "data-parallel step executes on RCCL; latencies are one-rank figures, not xGMI figures")
    if world == 1 and dist.is_initialized():
        from mico_amd import distributed as D
        D.force_dist(False)
        if reducer is not None:
            reducer.close()
        reducer = None
        dist.destroy_process_group()
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    samples = b * world * args.steps
    value = samples / elapsed
    kern = {0: "gemm_kernel<T,{ta},{tb},TileCfg<128,128,2,2,64,2>>", 1: "gemm_kernel<T,{ta},{tb},TileCfg<256,256,2,4,32,4>>",
            2: "gemm_pc_kernel<T,{ta},{tb},32>", 3: "gemm_w4_kernel<T,{ta},{tb}>", 4: "gemm_mx8_kernel<T> (fp8 e4m3 x E8M0/32, v_mfma_scale_f32_16x16x128)",
            5: "gemm_persist_kernel<T,{ta},{tb}> (256x256 8-wave ping-pong, persistent)",
            6: "gemm_kernel<T,{ta},{tb},TileCfg<256,128,2,2,32,3>> (two workgroups per CU)",
            7: "gemm_mid_kernel<T,{tb}> (256x128x64 unit ring, two workgroups per CU)"}
    role = {(0, 0): "y = x W^T (forward)", (0, 1): "dx = dy W", (1, 1): "dW = dy^T x", (1, 0): "x^T W"}

    def kname(key):
        ta, tb, kk = key
        if kk == 4:
            return kern[kk] + " : y = x W^T and dx = dy (W^T)^T"
        return kern[kk].format(ta=str(bool(ta)).lower(), tb=str(bool(tb)).lower()) + " : " + role[(ta, tb)]

    roofline = None
    if timer is not None:
        summ = timer.summary()
        per_variant = {kname(k): dict(launches=v["launches"], avg_ms=v["ms"] / v["launches"], tflops=v["flops"] / v["ms"] / 1e9)
                       for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        tot_flops = sum(v["flops"] for v in summ.values())
        tot_ms = sum(v["ms"] for v in summ.values())
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        achieved = dom[1]["flops"] / dom[1]["ms"] / 1e9
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from the separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (profiles/*_gemm_hbm_traffic.json, FETCH_SIZE
        # doubled as MI355X_MICROARCH.md prescribes for gfx950).  null when that profile is not present.
        traffic = None
        traffic_src = None
        for tname in ("r03_gemm_hbm_traffic.json", "r02_gemm_hbm_traffic.json", "r01_gemm_hbm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and world == 1:
                tj = json.load(open(tpath))
                key = {(0, 0): "NN", (0, 1): "dX", (1, 1): "dW", (1, 0): "TN"}[dom[0][:2]] + "_" + {0: "small", 1: "big", 2: "pc", 3: "w4", 4: "mx8", 5: "big", 6: "mid", 7: "mid"}[dom[0][2]]
                if key in tj:
                    traffic = tj[key]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + tname
                    break
        peak = 5000.0 if dom[0][2] == 4 else MFMA_PEAK_TFLOPS     # dense MX-fp8 peak (MI355X_MICROARCH.md) for the fp8 kernel
        roofline = dict(bound="mfma", kernel=kname(dom[0]), achieved=achieved, peak=peak, unit="TFLOP/s",
                        frac=achieved / peak, traffic=traffic,
                        traffic_unit="bytes/launch", traffic_provenance=(f"{traffic_src}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command "
                                                                          "on the builder's box (tools/profile_round.sh; FETCH_SIZE doubled for gfx950) - PMC counters "
                                                                          "cannot be read from inside this process, so this figure is NOT measured in this run"
                                                                          if traffic_src else None),
                        launches=dom[1]["launches"], avg_launch_ms=dom[1]["ms"] / dom[1]["launches"],
                        all_gemm=dict(tflops=tot_flops / tot_ms / 1e9, share_of_step_time=tot_ms / 1e3 / elapsed), variants=per_variant)
    workload = wl["key"]
    full = args.layers is None and args.vision == "evaclip01_giant" and args.task == wl["task"]
    # stochastic depth: a dropped (block, branch, frame) contributes exactly zero to values and gradients, so the engine does
    # not evaluate it.  The nominal (dense) FLOP count is what the reference executes; the executed count scales the ViT-block
    # share (frames x 40 blocks x 13.341 GF x 3 per sample) by the kept fraction of this run's draws.
    nominal = ALG_TFLOP_PER_SAMPLE.get(workload)
    vit_block_tf = wl["frames"] * 40 * 13.341 * 3 / 1e3
    executed = nominal - vit_block_tf * (1.0 - kept) if nominal else None
    # shared cross-attention K/V (runtime.CFG.share_cross_kv): the reference projects 4 condition sets per sample and layer (ITM triplet +
    # captioning pass, 1285 tokens each), the engine 2: 12 layers x 2 sets x 1285 x 2*768*1536 flop x 3 (fwd + dX + dW) = 0.218 TF/sample
    if executed is not None and runtime.CFG.share_cross_kv and args.task == "ret%tva_cap%tva" and not args.eval_mode:
        executed -= 12 * 2 * 1285 * 2 * 768 * 1536 * 3 / 1e12
    step_tflops = executed * value / world if (full and executed) else None
    res = {
        "metric": "omni-modal samples/sec (ViT-g/14 fwd+bwd)", "value": value, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype.split("-")[0], "data": "synthetic",
        "config": {"workload": (f"BASELINE.json configs[2]: ViT-g/14 image(1)+audio(4x224^2 mel windows)+text(77) fwd+bwd, "
                                f"b={b}/GPU, task {args.task} (ITC+ITM+CAP)" if args.workload == "img_aud_txt" else
                                f"BASELINE.json configs[3] per-rank share: ViT-g/14 image+video(9)+depth(1)+audio(4)+text(77) fwd+bwd, "
                                f"b={b}/GPU, task {args.task}" if args.workload == "omni" else
                                f"BASELINE.json configs[4] per-rank share: ViT-g/14 video (8 x 224^2 frames) + BERT cross-attention generative "
                                f"head (CAP), b={b}/GPU, task {args.task}, fp8 MFMA"),
                   "per_gpu_batch": b, "global_batch": b * world, "precision": PRECISIONS[args.dtype][3],
                   "vision": args.vision, "vit_layers": args.layers or "full", "parallelism": f"dp{world}",
                   "droppath": ("off (eval)" if args.eval_mode else "on, reference rates (0 -> 0.4 linear)"),
                   "droppath_schedule": ("dense: every branch evaluated then scaled by 0 | 1/keep" if args.dense_droppath
                                         else "dropped (block, branch, frame) triples are skipped - exact, zero contribution"),
                   "kept_branch_fraction": kept, "optimizer_step_in_timed_region": bool(args.optimizer),
                   "bert_dropout": (False if (args.eval_mode or args.no_bert_dropout) else
                                    "on: p=0.1 hidden + attention-probability (reference config.json)")},
        "samples_per_sec_per_gpu": value / world,
        "step_executed_tflops_per_gpu": step_tflops,
        "tflop_per_sample": {"dense_nominal": nominal, "executed": executed},
        "cross_kv": "condition K/V projected once per step and shared by the ITM triplet and the captioning pass" if runtime.CFG.share_cross_kv else "per pass",
        "step_mfma_frac": (step_tflops / MFMA_PEAK_TFLOPS) if step_tflops else None,
        "losses": {k: float(v.detach()) for k, v in losses.items()},
        "peak_mem_gb": peak_mem,
        "tower_plan": runtime.last_tower_plan,
        "roofline": roofline,
    }
    if comm is not None:
        res["comm"] = comm
    extras = not args.no_extras and args.workload == "img_aud_txt" and not args.optimizer
    if extras:
        # ---- the timed precision against the reference goldens, measured here and now ----
        res["parity"] = dict(precision=PRECISIONS[args.dtype][3], **measure_parity(model, dev))
    if extras and world == 1:
        # ---- the same step in the configuration with margin under the 1e-3 gate, timed next to it ----
        pc = "fp16-split-w" if args.dtype != "fp16-split-w" else "fp16"
        set_precision(pc)
        k2 = max(2, args.steps // 4)
        step()
        el2, _ = timed_steps(k2)
        res["parity_config"] = dict(precision=PRECISIONS[pc][3], value=b * k2 / el2, unit="samples/s", steps=k2, warmup=1,
                                    ms_per_step=el2 / k2 * 1e3, **{"parity": measure_parity(model, dev)})
        # ---- other precisions of the same step, for orientation (not gated configurations unless their parity says so) ----
        others = {}
        for oc in ("fp16-plain", "fp8"):
            if oc == args.dtype:
                continue
            try:
                set_precision(oc)
                step()
                elo, _ = timed_steps(k2)
                others[oc] = dict(precision=PRECISIONS[oc][3], value=b * k2 / elo, unit="samples/s", steps=k2, warmup=1, ms_per_step=elo / k2 * 1e3,
                                  parity=measure_parity(model, dev))
                if oc == "fp8":
                    others[oc]["parity_note"] = ("the golden inputs (1-2 images, <= 514 token rows) are below the size at which GEMMs route to the "
                                                 "fp8 kernel (>= 128 tiles of 256x256): this parity is the bf16 path's; the fp8 tolerance "
                                                 "(5.8-7.4e-2 forward) is measured by tests/test_model_gpu.py::test_fp8_tower_tolerance")
            except Exception as e:
                others[oc] = {"error": repr(e)}
        set_precision(args.dtype)
        res["other_precisions"] = others
        # ---- the metric's own multi-GPU configuration: one rank's share of BASELINE configs[3] (14 frames per sample: image + 8 video
        # frames + depth + 4 audio windows, b = 64 -> 896 tower frames), timed as a first-class object: >= 10 steps, its own executed-FLOP
        # figure, MFMA fraction and GEMM roofline, and the tower plan it ran under (functional.tower_plan: frames per pass, activation diet)
        try:
            wo = WORKLOADS["omni"]
            torch.cuda.empty_cache()
            ob = {k: v.to(dev) for k, v in synth_inputs(dict(b=b, **wo["shape"]), seed=4321).items()}
            step(ob, wo["task"])
            k3 = max(10, args.steps // 2)
            torch.cuda.reset_peak_memory_stats()
            otimer = None if args.no_gemm_timer else ops.KernelTimer()
            ops.GEMM_TIMER = otimer
            DropPlan.stats[:] = [0, 0]
            el3, _ = timed_steps(k3, ob, wo["task"])
            ops.GEMM_TIMER = None
            okept = DropPlan.stats[0] / DropPlan.stats[1] if DropPlan.stats[1] else 1.0
            onom = ALG_TFLOP_PER_SAMPLE[wo["key"]]
            oexec = onom - wo["frames"] * 40 * 13.341 * 3 / 1e3 * (1.0 - okept)
            if runtime.CFG.share_cross_kv and not args.eval_mode:
                # shared cross-attention K/V: the reference projects 4 condition sets for tva (ITM triplet + captioning pass, E = 13 x 257
                # tokens) and 3 for tvd (triplet only, E = 10 x 257), the engine 2 + 2: 12 layers x saved sets x E x 2*768*1536 flop x 3
                oexec -= 12 * (2 * 13 * 257 + 1 * 10 * 257) * 2 * 768 * 1536 * 3 / 1e12
            oval = b * k3 / el3
            oroof = None
            if otimer is not None:
                summ = otimer.summary()
                tf = sum(v["flops"] for v in summ.values())
                tm = sum(v["ms"] for v in summ.values())
                dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
                oroof = dict(bound="mfma", kernel=kname(dom[0]), achieved=dom[1]["flops"] / dom[1]["ms"] / 1e9, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=dom[1]["flops"] / dom[1]["ms"] / 1e9 / MFMA_PEAK_TFLOPS, launches=dom[1]["launches"],
                             avg_launch_ms=dom[1]["ms"] / dom[1]["launches"],
                             all_gemm=dict(tflops=tf / tm / 1e9, share_of_step_time=tm / 1e3 / el3))
            res["secondary"] = {"omni_configs3_rank_share": dict(
                value=oval, unit="samples/s", ms_per_step=el3 / k3 * 1e3, steps=k3, warmup=1, frames_per_sample=wo["frames"],
                frames_per_sec=oval * wo["frames"], task=wo["task"], peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30,
                precision=PRECISIONS[args.dtype][3], kept_branch_fraction=okept,
                tflop_per_sample={"dense_nominal": onom, "executed": oexec}, step_executed_tflops_per_gpu=oexec * oval,
                step_mfma_frac=oexec * oval / MFMA_PEAK_TFLOPS, roofline=oroof,
                tower_plan=runtime.last_tower_plan,
                tower_plan_note="frames_per_pass == frames: no chunked recompute; diet 1 / 2: MLP intermediates (and LayerNorm outputs) "
                                "recomputed in the backward (mico_amd.runtime.set_activation_diet)")}
            del ob
        except Exception as e:   # the headline line must survive a failure of the secondary measurement
            res["secondary"] = {"omni_configs3_rank_share": {"error": repr(e)}}
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(sd_cpu, args)
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
