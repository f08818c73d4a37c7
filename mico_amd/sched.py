"""Learning-rate multipliers of the reference trainer (data/utils/sched.py:3-29) as functions of training progress
x = global_step / num_train_steps: every schedule ramps linearly from 0 to 1 over the first `warmup_ratio` of training and then
follows its own tail.  Same public names and values as the reference (pinned in tests/golden/optimizer.pt)."""
import math


def _with_warmup(tail):
    def schedule(x, warmup_ratio):
        if x < warmup_ratio:
            return x / warmup_ratio
        return tail(x, warmup_ratio)
    return schedule


warmup_cosine = _with_warmup(lambda x, w: 0.5 * (1.0 + math.cos(math.pi * x)))          # half cosine over the whole run
warmup_constant = _with_warmup(lambda x, w: 1.0)                                        # flat after the ramp
warmup_linear = _with_warmup(lambda x, w: max((x - 1.0) / (w - 1.0), 0))                # triangular: back to 0 at x = 1

scheduler_dict = {"warmup_linear": warmup_linear, "warmup_cosine": warmup_cosine}


def get_lr_sched(global_step, opts):
    """multiplier for step `global_step` under opts.scheduler / opts.warmup_ratio / opts.num_train_steps (sched.py:25-29)"""
    progress = global_step / opts.num_train_steps
    return scheduler_dict[opts.scheduler](progress, opts.warmup_ratio)
