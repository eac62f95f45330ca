"""Learning-rate schedules of the reference training loops (host arithmetic only).

Drop-in for ``src/optimization/sched.py``: the loop calls ``get_lr_sched(global_step, decay, learning_rate,
num_train_steps, warmup_ratio=..., decay_epochs=..., multi_step_epoch=...)`` once per step
(``tasks/run_video_retrieval.py:372-376``) and writes the result into ``optimizer.param_groups`` (:377-383).

Every schedule is ``learning_rate * factor(step)`` with a linear ramp ``step / warmup`` first; they differ in the ramp's
end condition (``<`` for linear / cosine, ``<=`` for invsqrt / multi_step -- kept, the golden tables in
tests/golden/optim.pt pin both) and in the decay after it.  A non-positive result is replaced by 1e-8 (the reference's
guard against a miscounted ``num_train_steps``, sched.py:80-83).  ``warmup_linear`` & co. are exported under the
reference's names for callers that import them directly.
"""
import math
from typing import Callable, Dict, Sequence

LR_FLOOR = 1e-8


def _after_ramp(step: float, warmup: float, inclusive: bool, decay: Callable[[], float]) -> float:
    in_ramp = step <= warmup if inclusive else step < warmup
    return step / warmup if in_ramp else decay()


def warmup_linear(step, warmup_step, tot_step):
    """ramp, then a straight line down to 0 at ``tot_step`` (clamped at 0 beyond it)"""
    return _after_ramp(step, warmup_step, False, lambda: max(0, (tot_step - step) / (tot_step - warmup_step)))


def warmup_cosine(step, warmup_step, tot_step):
    """ramp, then half a cosine period from 1 to 0 over the remaining steps"""
    return _after_ramp(step, warmup_step, False,
                       lambda: 0.5 * (1.0 + math.cos(math.pi * (step - warmup_step) / (tot_step - warmup_step))))


def noam_schedule(step, warmup_step=4000):
    """ramp, then sqrt(warmup / step)"""
    return _after_ramp(step, warmup_step, True, lambda: math.sqrt(warmup_step) / math.sqrt(step))


def multi_step_schedule(n_epoch, milestones: Sequence[int], step, warmup_step, gamma=0.5):
    """ramp, then gamma^k with k = number of milestones already passed (gamma^(len+1) after the last one, as the
    reference has it)"""
    def decay():
        passed = sum(1 for m in milestones if n_epoch >= m)
        return gamma ** (passed if passed < len(milestones) else len(milestones) + 1)
    return _after_ramp(step, warmup_step, True, decay)


def get_lr_sched(global_step, decay, learning_rate, num_train_steps, warmup_ratio=0.1, decay_epochs=(),
                 multi_step_epoch=-1):
    warmup = int(warmup_ratio * num_train_steps)
    factors: Dict[str, Callable[[], float]] = {
        "linear": lambda: warmup_linear(global_step, warmup, num_train_steps),
        "cosine": lambda: warmup_cosine(global_step, warmup, num_train_steps),
        "invsqrt": lambda: noam_schedule(global_step, warmup),
        "constant": lambda: 1.0,
    }
    if decay == "multi_step":
        if multi_step_epoch < 0:
            raise AssertionError("multi_step needs multi_step_epoch >= 0")
        factor = multi_step_schedule(multi_step_epoch, list(decay_epochs), global_step, warmup)
    elif decay in factors:
        factor = factors[decay]()
    else:
        raise ValueError(f"unknown lr decay {decay!r}")      # the reference dies with an UnboundLocalError here
    lr = learning_rate * factor
    return lr if lr > 0 else LR_FLOOR
