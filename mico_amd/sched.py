"""Learning-rate schedules of the reference trainer (data/utils/sched.py:3-29): multipliers of the group's `init_lr` as a function
of training progress x = global_step / num_train_steps."""
import math


def warmup_cosine(x, warmup_ratio):
    return x / warmup_ratio if x < warmup_ratio else 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup_ratio):
    return x / warmup_ratio if x < warmup_ratio else 1.0


def warmup_linear(x, warmup_ratio):
    return x / warmup_ratio if x < warmup_ratio else max((x - 1.0) / (warmup_ratio - 1.0), 0)


scheduler_dict = {"warmup_linear": warmup_linear, "warmup_cosine": warmup_cosine}


def get_lr_sched(global_step, opts):
    return scheduler_dict[opts.scheduler](global_step / opts.num_train_steps, opts.warmup_ratio)
