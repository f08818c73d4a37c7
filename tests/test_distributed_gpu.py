"""Two ranks on ONE GPU (gloo process group carrying CUDA tensors): the complete MiCo.forward alignment step with the
packed all-gather, index-then-fetch negatives and the gradient reducer, checked against a single-process evaluation of the
same global batch.  Exercises everything of the N > 1 path except RCCL itself (a 1-GPU box cannot host two RCCL ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from common import build_model
        from mico_amd import runtime
        from mico_amd.weights import synth_inputs
        from mico_amd.distributed import GradBucketReducer
        dev = torch.device("cuda:0")
        runtime.set_compute_dtype(torch.float16)
        m, _ = build_model("evaclip02_base", 2, device=dev)
        b = 3
        inputs = [synth_inputs(dict(b=b, vision=1, audio=1, S=12), seed=50 + r) for r in range(world)]
        mine = {k: v.to(dev) for k, v in inputs[rank].items()}
        probe = {"tower block": m.vision_encoder.visual.blocks[1].attn.proj.weight,
                 "tower tail": m.vision_encoder.visual.patch_embed.proj.bias,
                 "bert": m.multimodal_encoder.bert.encoder.layer[3].output.dense.weight, "head": m.contra_head_va.weight}
        # fixed negatives (global indices) and token masks so both evaluations draw the same "random" numbers
        inj = {"tva": dict(neg_cond_idx=torch.tensor([(rank * b + i + 1) % (world * b) for i in range(b)]),
                           neg_text_idx=torch.tensor([(rank * b + i + 2) % (world * b) for i in range(b)]))}
        import random
        from mico_amd.model import TokenMasker
        mi, lab = TokenMasker(rng=random.Random(7 + rank))(inputs[rank]["input_ids"], 0.6)
        inj["cap"] = dict(masked_ids=mi, labels=lab)
        batch = dict(mine)
        batch["_injected"] = inj
        # pass 1 without the reducer: this rank's own gradients (the model is in eval mode: the step is deterministic)
        m.zero_grad(set_to_none=True)
        sum(m(dict(batch), "ret%tva_cap%tva").values()).backward()
        local = {k: p.grad.detach().float().clone() for k, p in probe.items()}
        # pass 2 with it: tower blocks are all-reduced from inside the tower backward (arena slices), the rest through buckets
        red = GradBucketReducer(m.parameters(), bucket_bytes=64 << 20)
        m.zero_grad(set_to_none=True)
        out = m(batch, "ret%tva_cap%tva")
        sum(out.values()).backward()
        assert len(red._early) > 10, "the tower's blocks must have been reduced through the arena-slice hook"
        red.finish()
        torch.cuda.synchronize()
        for k, p in probe.items():
            both = [torch.empty_like(local[k]) for _ in range(world)]
            dist.all_gather(both, local[k])
            mean = (both[0] + both[1]) / 2
            err = (p.grad.float() - mean).abs().max() / mean.abs().max().clamp_min(1e-20)
            assert err < 1e-5, (k, float(err))
        res = {k: float(v) for k, v in out.items()}
        g = m.contra_head_va.weight.grad.detach().float().cpu().clone()
        g2 = m.vision_encoder.visual.blocks[0].mlp.w1.weight.grad.detach().float().cpu().clone()
        # ---- single-process reference of THIS rank's loss on the global batch (simulated world) ----
        if rank == 0:
            with torch.no_grad():
                encs = [m.encode_batch({k: v.to(dev) for k, v in inp.items()}) for inp in inputs]
            world_d = dict(rank=0, feat_t_all=torch.cat([e["feat_t"] for e in encs]),
                           ids_all=torch.cat([i["input_ids"] for i in inputs]).to(dev),
                           mask_all=torch.cat([i["attention_mask"] for i in inputs]).to(dev))
            world_d["feat_va_all"] = torch.cat([m._feat_cond(e, "va") for e in encs])
            remote = torch.cat([m._condition_feats(e, "va") for e in encs[1:]]).detach()
            world_d["cond_va_fetch"] = lambda cond, idx: torch.cat((cond, remote))[idx]
            b2 = dict(mine)
            b2["_injected"] = inj
            b2["_world"] = world_d
            ref = m(b2, "ret%tva_cap%tva")
            for k in ref:
                assert abs(float(ref[k]) - res[k]) < 2e-3 * max(1.0, abs(float(ref[k]))), (k, float(ref[k]), res[k])
        # averaged gradients must be identical on both ranks
        gs = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        assert torch.equal(gs[0], gs[1])
        gs2 = [torch.empty_like(g2) for _ in range(world)]
        dist.all_gather(gs2, g2)
        assert torch.equal(gs2[0], gs2[1]) and gs2[0].abs().max() > 0
        ret[rank] = "ok"
    except Exception:   # noqa
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(cuda):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in range(2):
        assert ret.get(r) == "ok", ret.get(r)
