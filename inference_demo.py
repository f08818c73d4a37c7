#!/usr/bin/env python
"""inference_demo.py - the reference's demo entry point (inference_demo.py:14-174) on the MI355X path.

    python inference_demo.py --pretrain_dir MiCo-g --image example/test.jpeg            # a real checkpoint directory
    python inference_demo.py --synthetic evaclip01_giant --image some.jpeg               # no checkpoint: synthetic weights

`load_from_pretrained_dir(pretrain_dir, video_resolution, return_modal) -> (checkpoint, model_cfg)` keeps the reference's
contract: reads log/hps.json, picks ckpt/model_step_<max>.pt, renames video->vision / evaclip_model|clip_model->vision_encoder,
casts to fp32, nearest-interpolates the frame embeddings to max_*_sample_num and bilinearly interpolates the ViT position
table to the requested resolution.  The demo then encodes the image and the texts, prints the text-to-image similarity, the ITM
scores and a beam-search caption (BertForMaskedLM.generate, :161-174).
"""
import argparse
import json
import os
from collections import defaultdict

import torch
import torch.nn.functional as F

from mico_amd.model import MiCo, AttrDict, default_cfg


def load_from_pretrained_dir(pretrain_dir, video_resolution=224, return_modal="full"):
    checkpoint_dir = os.path.join(pretrain_dir, "ckpt")
    file_cfg = json.load(open(os.path.join(pretrain_dir, "log", "hps.json")))
    model_cfg = AttrDict(file_cfg["model_cfg"])
    steps = sorted(int(i.split("_")[2].split(".")[0]) for i in os.listdir(checkpoint_dir) if i.startswith("model_step"))
    ckpt_file = os.path.join(checkpoint_dir, f"model_step_{steps[-1]}.pt")
    checkpoint = torch.load(ckpt_file, map_location="cpu")
    print(f"load_from_pretrained: {ckpt_file}")
    new_ckpt = {}
    for k, v in checkpoint.items():
        if "video" in k:
            new_ckpt[k.replace("video", "vision")] = v
        elif "evaclip_model" in k:
            new_ckpt[k.replace("evaclip_model", "vision_encoder")] = v
        elif "clip_model" in k:
            new_ckpt[k.replace("clip_model", "vision_encoder")] = v
        else:
            new_ckpt[k] = v.float()
    checkpoint = new_ckpt
    if model_cfg.frame_embedding_type == "adaptive":
        vkey = "vision_frame_embedding" if "vision_frame_embedding" in checkpoint else "vision_perceiver.vision_frame_embedding"
        wanted = [(vkey, model_cfg.max_vision_sample_num)]      # a checkpoint with neither vision key is a KeyError, as upstream
        if "audio_frame_embedding" in checkpoint:
            wanted.append(("audio_frame_embedding", model_cfg.max_audio_sample_num))
        for key, n in wanted:
            emb = checkpoint[key]
            if emb.shape[1] != n:
                checkpoint[key] = F.interpolate(emb.permute(0, 2, 1), n, mode="nearest").permute(0, 2, 1)
    vtype = model_cfg.vision_encoder_type
    if vtype.startswith("clip") or vtype.startswith("evaclip"):
        # OpenAI-CLIP towers keep a 2-D table, EVA towers a [1, 1+g*g, D] one; both resize the patch part bilinearly
        eva = vtype.startswith("evaclip")
        pk = "vision_encoder.visual.pos_embed" if eva else "vision_encoder.visual.positional_embedding"
        wk = "vision_encoder.visual.patch_embed.proj.weight" if eva else "vision_encoder.visual.conv1.weight"
        table = checkpoint[pk][0] if eva else checkpoint[pk]
        width, patch = table.shape[-1], checkpoint[wk].shape[-1]
        grid = round((table.shape[0] - 1) ** 0.5)
        new_grid = model_cfg.vision_resolution // patch
        if new_grid != grid:
            oth = table[1:].reshape(grid, grid, width).permute(2, 0, 1).unsqueeze(0)
            oth = F.interpolate(oth, (new_grid, new_grid), mode="bilinear")[0].permute(1, 2, 0).reshape(-1, width)
            table = torch.cat((table[0:1], oth), dim=0)
            checkpoint[pk] = table.unsqueeze(0) if eva else table
    if return_modal == "uni":
        out = defaultdict()
        for k in checkpoint:
            if "video_encoder" in k:
                out[".".join(k.split(".")[1:])] = checkpoint[k]
        checkpoint = out
    elif return_modal == "text":
        out = defaultdict()
        for k in checkpoint:
            if "multimodal_encoder" in k:
                out[".".join(k.split(".")[1:])] = checkpoint[k]
        checkpoint = out
    return checkpoint, model_cfg


def write_synthetic_pretrain_dir(path, vision_encoder_type="evaclip01_giant", steps=(5, 10), seed=0, **cfg_over):
    """A stand-in for the released `MiCo-g/` directory (no network here): hps.json + ckpt/model_step_<n>.pt with synthetic
    weights stored under the *pre-rename* key names a released checkpoint uses, so the loader's remap path is exercised."""
    from mico_amd.weights import synth_state_dict
    cfg = default_cfg(vision_encoder_type, **cfg_over)
    os.makedirs(os.path.join(path, "ckpt"), exist_ok=True)
    os.makedirs(os.path.join(path, "log"), exist_ok=True)
    json.dump({"model_cfg": dict(cfg)}, open(os.path.join(path, "log", "hps.json"), "w"))
    m = MiCo(cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    stored = {}
    for k, v in sd.items():
        k2 = k.replace("vision_encoder", "evaclip_model") if k.startswith("vision_encoder") else k.replace("vision_", "video_")
        stored[k2] = v.clone()
    for s in steps:
        torch.save(stored if s == max(steps) else {}, os.path.join(path, "ckpt", f"model_step_{s}.pt"))
    return cfg, sd


@torch.no_grad()
def run_demo(model, image_input, texts, device="cuda", max_length=30):
    """The retrieval + matching part of the reference's __main__ (inference_demo.py:128-158)."""
    image_input = image_input.to(device).unsqueeze(1)          # image as a 1 frame video
    video_output = model.forward_vision_encoder(image_input)
    feat_v = F.normalize(model.contra_head_v(model.pool_vision_for_contra(video_output)), dim=-1)
    tok = model.multimodal_encoder.tokenizer(texts, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt")
    input_ids, attention_mask = tok.input_ids.to(device), tok.attention_mask.to(device)
    caption_output = model.forward_multimodal_encoder(input_ids, attention_mask).sequence_output
    feat_t = F.normalize(model.contra_head_t(model.pool_text_for_contra(caption_output)), dim=-1)
    sim_t2v = torch.matmul(feat_t, feat_v.permute(1, 0))
    video_input = model.get_multimodal_forward_input_vision(video_output)
    video_input = video_input.expand(input_ids.shape[0], -1, -1).contiguous()
    slice_output = model.forward_multimodal_encoder(input_ids, attention_mask, video_input).sequence_output
    slice_scores = F.softmax(model.itm_head(slice_output[:, 0]), dim=1)[:, 1]
    # caption generation (inference_demo.py:161-174)
    cap_input = model.get_multimodal_forward_input_vision(video_output)
    tk = model.multimodal_encoder.tokenizer
    init_ids = torch.full((cap_input.size(0), 1), tk.bos_token_id, dtype=torch.long, device=device)
    outputs = model.multimodal_encoder.generate(input_ids=init_ids, attention_mask=init_ids.new_ones(cap_input.size(0), 1, 1),
                                                encoder_hidden_states=cap_input, max_new_tokens=model.max_caption_len,
                                                num_beams=model.beam_size, eos_token_id=tk.sep_token_id,
                                                pad_token_id=tk.pad_token_id, length_penalty=0.6)
    captions = tk.batch_decode(outputs[:, 1:], skip_special_tokens=True)
    return dict(feat_v=feat_v, feat_t=feat_t, sim_t2v=sim_t2v, itm_scores=slice_scores, input_ids=input_ids,
                caption_ids=outputs, captions=captions)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrain_dir", default="MiCo-g")
    ap.add_argument("--synthetic", default=None, help="vision_encoder_type: build a synthetic pretrain dir instead of reading one")
    ap.add_argument("--image", default="example/test.jpeg")
    ap.add_argument("--texts", nargs="*", default=["a man is skiing in a snowy day.", "it's a hot day"])
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    args = ap.parse_args()
    device = "cuda"
    from mico_amd import runtime
    from mico_amd.model.imageprocessor import ImageProcessor
    runtime.set_compute_dtype(torch.float16 if args.dtype == "fp16" else torch.bfloat16)
    if args.synthetic:
        import tempfile
        args.pretrain_dir = tempfile.mkdtemp(prefix="mico_synth_")
        write_synthetic_pretrain_dir(args.pretrain_dir, args.synthetic)
    checkpoint, opts = load_from_pretrained_dir(args.pretrain_dir, video_resolution=224, return_modal="full")
    model = MiCo.from_pretrained(opts, checkpoint).to(device).eval()
    proc = ImageProcessor(image_resolution=224, image_encoder_type="swin", training=True)
    image_input = proc(args.image)
    if image_input is None:
        raise SystemExit(f"cannot read {args.image}")
    out = run_demo(model, image_input, args.texts, device)
    print(out["sim_t2v"])
    print(out["itm_scores"])
    print(out["captions"])


if __name__ == "__main__":
    main()
