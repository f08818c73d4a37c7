#!/usr/bin/env python
"""bench.py - MiCo omni-modal alignment step (ViT-g/14 fwd+bwd) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the single-GPU ViT-g/14 configuration the metric is quoted on): per GPU b = 64 samples
of image (1 frame, 224^2) + audio (4 spectrogram windows of 224x224 = 10 s of mel-spec) + text (77 tokens), synthetic
inputs resident in HBM, random-init weights of the real architecture (EVA01-CLIP-g/14 tower shared by image and audio,
BERT-base with cross-attention), one step = forward + backward of the full alignment loss "ret%tva_cap%tva" (ITC + ITM with
in-batch hard negatives + causal masked-caption LM; step-B of SURVEY.md section 8d), bf16 MFMA with fp32 accumulation /
residual stream / statistics.  For N > 1 every rank runs that workload on its own shard (weak scaling), the contrastive
features are exchanged with one packed RCCL all-gather, hard-negative condition rows with an index-then-fetch all-to-all,
and gradients are averaged with bucketed all-reduces overlapped with backward.

One JSON line on rank 0 (see the repo prompt for the field contract) with two extra objects:
  roofline     - the dominant kernel (MFMA GEMM) timed per launch with HIP events inside the timed region;
  cpu_baseline - the CPU oracle (oracle/mico_oracle.py, a restatement of the reference parity-locked to it) timed on the
                 host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic forward GF / sample for step-B at config-3 shapes: 2 * MAC of every
# GEMM incl. attention, LM head only where a loss consumes it; fwd+bwd = 3 x fwd
ALG_TFLOP_PER_SAMPLE = {"vitg_img1_aud4_txt77_stepB": 8.74}
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--vision", default="evaclip01_giant")
    ap.add_argument("--layers", type=int, default=None, help="truncate the ViT (debug only; invalidates the metric)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--task", default=None)
    ap.add_argument("--workload", default="img_aud_txt", choices=["img_aud_txt", "omni"],
                    help="img_aud_txt = BASELINE configs[2] (the metric's single-GPU configuration, default); omni = one rank's share "
                         "of configs[3]: image+video (9 vision frames) + depth + audio + text, 14 frames/sample (not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--eval-mode", action="store_true", help="disable DropPath (parity-style run)")
    ap.add_argument("--gemm-detail", action="store_true", help="print a per-shape table of the timed GEMM launches to stderr")
    ap.add_argument("--optimizer", action="store_true",
                    help="also run the AdamW step (mico_amd.optim, SURVEY section 8 row f4) inside the timed step: a full training step, "
                         "beyond the metric's fwd+bwd definition")
    ap.add_argument("--no-bert-dropout", action="store_true", help="A/B switch: BERT dropout probabilities set to 0 (invalidates the metric)")
    ap.add_argument("--dense-droppath", action="store_true",
                    help="evaluate dropped residual branches too and multiply them by 0 (the reference's schedule) instead of skipping them")
    return ap.parse_args()


def cpu_baseline(sd_cpu, args):
    """Oracle step-B forward+backward on the host cores, bounded sample: --cpu-batch samples, ONE untimed-warmup-free step."""
    from oracle import mico_oracle as O
    from mico_amd.weights import synth_inputs
    import random
    # 32 threads: PyTorch's CPU GEMMs on this path stop scaling (and regress badly) far below the 256 hardware threads of the
    # GPU box's host - a first run with all 256 threads took 1292 s for the same sample.
    ncores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    arch = O.ARCHS[args.vision]
    b = args.cpu_batch
    inp = synth_inputs(dict(b=b, **WORKLOADS[args.workload]["shape"]), seed=99)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd_cpu.items()}
    sd["multimodal_encoder.cls.predictions.decoder.weight"] = sd["multimodal_encoder.bert.embeddings.word_embeddings.weight"]
    mi, lab = O.token_masker(inp["input_ids"], 0.6, random.Random(0))
    idx = torch.arange(b).roll(1)
    injected = {"tva": dict(neg_cond_idx=idx, neg_text_idx=idx), "cap": dict(masked_ids=mi, labels=lab)}
    t0 = time.time()
    out, _ = O.mico_forward(sd, arch, inp, args.task, dict(itm_ratio=0.1), injected=injected)
    sum(out.values()).backward()
    dt = time.time() - t0
    return dict(value=b / dt, unit="samples/s", cores=ncores, kind="port",
                sample=f"oracle/mico_oracle.py fp32, same step ({args.task}) at b={b} (image 1 + audio 4 frames + 77 tokens), "
                       f"1 step, {dt:.1f} s on {ncores} threads")


WORKLOADS = {
    "img_aud_txt": dict(shape=dict(vision=1, audio=4, S=77), task="ret%tva_cap%tva", key="vitg_img1_aud4_txt77_stepB", frames=5),
    "omni": dict(shape=dict(vision=9, depth=1, audio=4, S=77), task="ret%tva%tvd_cap%tva", key="vitg_omni14_txt77", frames=14),
}


def main():
    args = parse()
    wl = WORKLOADS[args.workload]
    args.task = args.task or wl["task"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from mico_amd import runtime, ops
    from mico_amd.model import MiCo, default_cfg
    from mico_amd.weights import synth_state_dict, synth_inputs
    from mico_amd.distributed import GradBucketReducer

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    runtime.set_compute_dtype(dtype)
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

    def step():
        model.zero_grad(set_to_none=True)
        losses = model(dict(batch), args.task, compute_loss=True)
        total = sum(losses.values())
        total.backward()
        if reducer is not None:
            reducer.finish()
        if optimizer is not None:
            optimizer.step()
        return losses

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timer = ops.KernelTimer()
    ops.GEMM_TIMER = timer
    DropPlan.stats[:] = [0, 0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.GEMM_TIMER = None
    if args.gemm_detail and rank == 0:
        import sys
        tab = {}
        for (var, flops, e0, e1), det in zip(timer.records, timer.detail):
            d = tab.setdefault((var, det[1], det[2], det[3], det[4]), [0, 0.0, 0.0, 0])
            d[0] += 1; d[1] += flops; d[2] += e0.elapsed_time(e1); d[3] += det[0]
        print(f"{'(ta,tb,kernel)':>14s} {'N':>6s} {'K':>6s} {'epilogue':>10s} {'split':>5s} {'launches':>8s} {'avg M':>8s} {'avg us':>8s} {'TFLOP/s':>8s} {'ms/step':>8s}", file=sys.stderr)
        for key, d in sorted(tab.items(), key=lambda kv: -kv[1][2]):
            print(f"{str(key[0]):>14s} {key[1]:6d} {key[2]:6d} {key[3]:>10s} {key[4]:5d} {d[0]:8d} {d[3] / d[0]:8.0f} {d[2] / d[0] * 1e3:8.1f} "
                  f"{d[1] / d[2] / 1e9:8.1f} {d[2] / args.steps:8.2f}", file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    samples = b * world * args.steps
    value = samples / elapsed
    summ = timer.summary()
    kern = {0: "gemm_kernel<T,{ta},{tb},TileCfg<128,128,2,2,64,2>>", 1: "gemm_kernel<T,{ta},{tb},TileCfg<256,256,2,4,32,4>>",
            2: "gemm_pc_kernel<T,{ta},{tb},32>"}
    role = {(0, 0): "y = x W^T (forward)", (0, 1): "dx = dy W", (1, 1): "dW = dy^T x", (1, 0): "x^T W"}

    def kname(key):
        ta, tb, kk = key
        return kern[kk].format(ta=str(bool(ta)).lower(), tb=str(bool(tb)).lower()) + " : " + role[(ta, tb)]

    per_variant = {kname(k): dict(launches=v["launches"], avg_ms=v["ms"] / v["launches"], tflops=v["flops"] / v["ms"] / 1e9)
                   for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
    tot_flops = sum(v["flops"] for v in summ.values())
    tot_ms = sum(v["ms"] for v in summ.values())
    dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
    achieved = dom[1]["flops"] / dom[1]["ms"] / 1e9
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from the separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (profiles/r01_gemm_hbm_traffic.json, FETCH_SIZE
    # doubled as MI355X_MICROARCH.md prescribes for gfx950).  null when that profile is not present.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic.json")
    if os.path.exists(tpath) and world == 1:
        tj = json.load(open(tpath))
        key = {(0, 0): "NN", (0, 1): "dX", (1, 1): "dW", (1, 0): "TN"}[dom[0][:2]] + "_" + {0: "small", 1: "big", 2: "pc"}[dom[0][2]]
        if key in tj:
            traffic = tj[key]["hbm_bytes_per_launch"]
    roofline = dict(bound="mfma", kernel=kname(dom[0]), achieved=achieved, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=achieved / MFMA_PEAK_TFLOPS, traffic=traffic, traffic_unit="bytes/launch (PMC pass, see profiles/)",
                    launches=dom[1]["launches"],
                    avg_launch_ms=dom[1]["ms"] / dom[1]["launches"],
                    all_gemm=dict(tflops=tot_flops / tot_ms / 1e9, share_of_step_time=tot_ms / 1e3 / elapsed), variants=per_variant)
    workload = wl["key"]
    full = args.layers is None and args.vision == "evaclip01_giant" and args.task == "ret%tva_cap%tva" and args.workload == "img_aud_txt"
    # stochastic depth: a dropped (block, branch, frame) contributes exactly zero to values and gradients, so the engine does
    # not evaluate it.  The nominal (dense) FLOP count is what the reference executes; the executed count scales the ViT-block
    # share (5 frames x 40 blocks x 13.341 GF x 3 = 8.00 of the 8.74 TF/sample) by the kept fraction of this run's draws.
    kept = DropPlan.stats[0] / DropPlan.stats[1] if DropPlan.stats[1] else 1.0
    nominal = ALG_TFLOP_PER_SAMPLE.get(workload)
    executed = nominal - 8.00 * (1.0 - kept) if nominal else None
    # shared cross-attention K/V (runtime.CFG.share_cross_kv): the reference projects 4 condition sets per sample and layer (ITM triplet +
    # captioning pass, 1285 tokens each), the engine 2: 12 layers x 2 sets x 1285 x 2*768*1536 flop x 3 (fwd + dX + dW) = 0.218 TF/sample
    from mico_amd import runtime as _rt
    if executed is not None and _rt.CFG.share_cross_kv and args.task == "ret%tva_cap%tva" and not args.eval_mode:
        executed -= 12 * 2 * 1285 * 2 * 768 * 1536 * 3 / 1e12
    step_tflops = executed * value / world if full else None
    res = {
        "metric": "omni-modal samples/sec (ViT-g/14 fwd+bwd)", "value": value, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"BASELINE.json configs[2]: ViT-g/14 image(1)+audio(4x224^2 mel windows)+text(77) fwd+bwd, "
                                f"b={b}/GPU, task {args.task} (ITC+ITM+CAP)" if args.workload == "img_aud_txt" else
                                f"BASELINE.json configs[3] per-rank share: ViT-g/14 image+video(9)+depth(1)+audio(4)+text(77) fwd+bwd, "
                                f"b={b}/GPU, task {args.task}; tower chunked with recompute"),
                   "per_gpu_batch": b, "global_batch": b * world,
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
        "cross_kv": "condition K/V projected once per step and shared by the ITM triplet and the captioning pass" if _rt.CFG.share_cross_kv else "per pass",
        "step_mfma_frac": (step_tflops / MFMA_PEAK_TFLOPS) if step_tflops else None,
        "losses": {k: float(v.detach()) for k, v in losses.items()},
        "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(sd_cpu, args)
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
