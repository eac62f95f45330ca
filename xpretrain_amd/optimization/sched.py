"""Learning-rate schedules of the reference training loops (host arithmetic only).

Mirrors ``src/optimization/sched.py``: ``noam_schedule`` :8-11, ``warmup_linear`` :14-17, ``warmup_cosine`` :19-23,
``multi_step_schedule`` :25-33, ``get_lr_sched`` :62-84 -- same names, arguments and corner cases (the ``<= 0 -> 1e-8``
guard), so ``run_video_retrieval.py:372-383`` can import this module instead.
"""
import math


def noam_schedule(step, warmup_step=4000):
    if step <= warmup_step:
        return step / warmup_step
    return (warmup_step ** 0.5) * (step ** -0.5)


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def warmup_cosine(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    progress = (step - warmup_step) / (tot_step - warmup_step)
    return 0.5 * (1.0 + math.cos(math.pi * progress))


def multi_step_schedule(n_epoch, milestones, step, warmup_step, gamma=0.5):
    if step <= warmup_step:
        return step / warmup_step
    for i, m in enumerate(sorted(milestones)):
        if n_epoch < m:
            return gamma ** i
    return gamma ** (len(milestones) + 1)


def get_lr_sched(global_step, decay, learning_rate, num_train_steps, warmup_ratio=0.1, decay_epochs=(),
                 multi_step_epoch=-1):
    warmup_steps = int(warmup_ratio * num_train_steps)
    if decay == "linear":
        lr = learning_rate * warmup_linear(global_step, warmup_steps, num_train_steps)
    elif decay == "cosine":
        lr = learning_rate * warmup_cosine(global_step, warmup_steps, num_train_steps)
    elif decay == "invsqrt":
        lr = learning_rate * noam_schedule(global_step, warmup_steps)
    elif decay == "constant":
        lr = learning_rate
    elif decay == "multi_step":
        assert multi_step_epoch >= 0
        lr = learning_rate * multi_step_schedule(multi_step_epoch, list(decay_epochs), global_step, warmup_steps)
    else:
        raise ValueError(f"unknown lr decay {decay!r}")     # the reference hits an UnboundLocalError here
    return lr if lr > 0 else 1e-8
