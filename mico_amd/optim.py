"""Optimizer step of the reference trainer on the MI355X.

`AdamW` keeps the constructor, state keys (`step`, `exp_avg`, `exp_avg_sq`) and arithmetic of data/utils/build_optimizer.py:105-197
(decoupled weight decay applied after the Adam update, eps 1e-6 added to sqrt(v), optional bias correction), so optimizer
checkpoints interchange.  `step()` is ONE kernel launch per (parameter group, step count): a multi-tensor pass over a descriptor
table (mico_adamw_step), which also refreshes the 16-bit GEMM-operand mirrors of the weights the engine caches
(runtime.gemm_weight) - no re-cast pass follows an optimizer step.  `build_optimizer` reproduces the reference's grouping
(:11-76): basic / new / CLIP-visual parameters, each with and without weight decay."""
import ctypes as C
import math

import torch

from . import _lib, runtime

CHUNK = 1 << 16


class _Desc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("numel", C.c_int64),
                ("w16", C.c_void_p), ("ld16", C.c_int64), ("lo_off", C.c_int64), ("cols", C.c_int), ("w16_dtype", C.c_int)]


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._chunk_cache = {}

    def _chunks(self, numels, dev):
        key = (tuple(numels), str(dev))
        hit = self._chunk_cache.get(key)
        if hit is None:
            ct, cs = [], []
            for t, n in enumerate(numels):
                for s in range(0, n, CHUNK):
                    ct.append(t)
                    cs.append(s)
            hit = (torch.tensor(ct, dtype=torch.int32).to(dev), torch.tensor(cs, dtype=torch.int64).to(dev), len(ct))
            self._chunk_cache[key] = hit
        return hit

    def _grad_table(self, plist):
        """descriptor table (gradients only) + chunk lists over `plist`, for mico_grads_finite"""
        dev = plist[0].device
        descs = (_Desc * len(plist))()
        keep = []
        for d, p in zip(descs, plist):
            g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
            keep.append(g)
            d.g, d.numel = g.data_ptr(), p.numel()
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        return table, keep, self._chunks([p.numel() for p in plist], dev)

    @torch.no_grad()
    def grads_nonfinite(self, flag):
        """flag (fp32 device scalar) <- 1 if any gradient this optimizer would consume is inf / NaN; no host sync here."""
        plist = [p for group in self.param_groups for p in group["params"] if p.grad is not None]
        if not plist:
            return flag
        table, keep, (ct, cs, n) = self._grad_table(plist)
        _lib.check(_lib.lib().mico_grads_finite(table.data_ptr(), len(plist), ct.data_ptr(), cs.data_ptr(), n, CHUNK, flag.data_ptr(),
                                                torch.cuda.current_stream(plist[0].device).cuda_stream), "mico_grads_finite")
        del keep
        return flag

    @torch.no_grad()
    def step(self, closure=None, grad_mult=1.0):
        """grad_mult: gradients are multiplied by it inside the update kernel (GradScaler: 1 / loss scale)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        refreshed = set()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if not p.is_cuda:
                    raise _lib.MicoHipError("mico_amd.optim.AdamW updates device parameters only (no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros_like(p.data, dtype=torch.float32)
                st["step"] += 1
                by_step.setdefault(st["step"], []).append(p)
            for step, plist in by_step.items():
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
                dev = plist[0].device
                descs = (_Desc * len(plist))()
                keep = []
                for d, p in zip(descs, plist):
                    st = self.state[p]
                    g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
                    keep.append(g)
                    assert p.data.is_contiguous() and p.dtype == torch.float32
                    d.p, d.g, d.m, d.v, d.numel = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                    mir = runtime.weight_mirror(p)
                    if mir is not None:
                        d.w16, d.ld16, d.lo_off, d.cols, d.w16_dtype = mir
                        refreshed.add(runtime.param_uid(p))
                    else:
                        d.w16, d.ld16, d.lo_off, d.cols, d.w16_dtype = None, 0, 0, 1, 0
                table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
                ct, cs, n = self._chunks([p.numel() for p in plist], dev)
                rc = lib.mico_adamw_step(table.data_ptr(), len(plist), ct.data_ptr(), cs.data_ptr(), n, CHUNK, float(group["lr"]),
                                         float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), float(step_size),
                                         float(grad_mult), torch.cuda.current_stream(dev).cuda_stream)
                _lib.check(rc, "mico_adamw_step")
                del keep
        runtime.after_optimizer_step(refreshed)
        return loss


class GradScaler:
    """Dynamic loss scaling with the interface and semantics of torch.cuda.amp.GradScaler as the reference trainer uses it
    (data/utils/pipeline.py:30,88,106-107: scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()): the loss is
    multiplied by the scale, step() checks the gradients for inf / NaN and SKIPS the optimizer step when it finds any (one device ->
    host read of a flag, exactly where torch's scaler has its .item()), update() halves the scale after a skipped step and doubles it
    after growth_interval clean ones.  On the MI355X path the un-scaling is not a pass of its own: mico_adamw_step multiplies the
    gradients by 1 / scale while it reads them, and the overflow check is one read-only multi-tensor kernel (mico_grads_finite).
    (The engine additionally carries its 16-bit activations' gradients x4096 inside each fp16 function; that internal scale never
    reaches a parameter gradient and is independent of this one.)"""

    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._scale, self._growth, self._backoff, self._interval = float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval)
        self._good_steps, self._enabled, self._found_inf = 0, enabled, None

    def scale(self, loss):
        return loss * self._scale if self._enabled else loss

    def get_scale(self):
        return self._scale

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if not hasattr(optimizer, "grads_nonfinite"):
            raise TypeError("mico_amd.optim.GradScaler drives mico_amd.optim.AdamW (fused un-scale + overflow check)")
        dev = next(p for g in optimizer.param_groups for p in g["params"]).device
        flag = torch.zeros(1, dtype=torch.float32, device=dev)
        optimizer.grads_nonfinite(flag)
        self._found_inf = bool(flag.item())      # the one host sync of the step (torch's GradScaler.step has the same one)
        if self._found_inf:
            return None
        return optimizer.step(*args, grad_mult=1.0 / self._scale, **kwargs)

    def update(self, new_scale=None):
        if not self._enabled:
            return
        if new_scale is not None:
            self._scale, self._good_steps = float(new_scale), 0
        elif self._found_inf is None:
            raise RuntimeError("GradScaler.update() before step(): no inf / nan check was recorded for this iteration")
        elif self._found_inf:
            self._scale, self._good_steps = self._scale * self._backoff, 0
        else:
            self._good_steps += 1
            if self._good_steps == self._interval:
                self._scale, self._good_steps = self._scale * self._growth, 0
        self._found_inf = None

    def state_dict(self):
        return {"scale": self._scale, "growth_factor": self._growth, "backoff_factor": self._backoff, "growth_interval": self._interval,
                "_growth_tracker": self._good_steps} if self._enabled else {}

    def load_state_dict(self, sd):
        if sd:
            self._scale, self._growth, self._backoff = float(sd["scale"]), float(sd["growth_factor"]), float(sd["backoff_factor"])
            self._interval, self._good_steps = int(sd["growth_interval"]), int(sd["_growth_tracker"])


def build_optimizer(model, args, checkpoint_optim=None):
    """data/utils/build_optimizer.py:11-93.  args.run_cfg: learning_rate, new_lr, clip_lr, weight_decay, betas, optim,
    new_params_name; args.model_cfg.vision_encoder_type."""
    vision_clip = "vision_encoder_type" in args.model_cfg and "clip" in args.model_cfg.vision_encoder_type
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    buckets = {k: [] for k in ("basic", "basic_nd", "new", "new_nd", "clip", "clip_nd")}
    names = {k: [] for k in ("basic", "new", "clip")}
    for k, v in model.named_parameters():
        nd = any(n in k for n in no_decay)
        if any(n in k for n in args.run_cfg.new_params_name):
            kind = "new"
        elif vision_clip and "visual" in k:
            kind = "clip"
        else:
            kind = "basic"
        buckets[kind + ("_nd" if nd else "")].append(v)
        names[kind].append(k)
    rc = args.run_cfg
    groups = [
        {"params": buckets["basic"], "weight_decay": rc.weight_decay, "lr": rc.learning_rate},
        {"params": buckets["basic_nd"], "weight_decay": 0.0, "lr": rc.learning_rate},
        {"params": buckets["new"], "weight_decay": rc.weight_decay, "lr": rc.new_lr},
        {"params": buckets["new_nd"], "weight_decay": 0.0, "lr": rc.new_lr},
        {"params": buckets["clip"], "weight_decay": rc.weight_decay, "lr": rc.clip_lr},
        {"params": buckets["clip_nd"], "weight_decay": 0.0, "lr": rc.clip_lr},
    ]
    if rc.optim != "adamw":
        raise ValueError("invalid optimizer" if rc.optim not in ("adam", "adamax") else
                         f"optimizer '{rc.optim}' is not provided by mico_amd (the MiCo/VAST configs use adamw)")
    for g in groups:
        g["init_lr"] = g["lr"]
    optimizer = AdamW(groups, lr=rc.learning_rate, betas=rc.betas)
    optimizer.new_params_name = names["new"]
    optimizer.new_lr = rc.new_lr
    optimizer.basic_lr = rc.learning_rate
    optimizer.clip_lr_visual = rc.clip_lr
    optimizer.clip_lr_visual_len = len(buckets["clip"])
    optimizer.zero_grad()
    if checkpoint_optim:
        optimizer.load_state_dict(checkpoint_optim)
    return optimizer
