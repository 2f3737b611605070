"""Warmup-Stable-Decay learning-rate schedule, host side of the optimizer step.

Mirror of the reference's ``src/scheduler.py`` (``WarmupStableDecayLR`` :11-94, ``create_scheduler``
:97-159): same constructor arguments, attribute names (so ``state_dict()`` round-trips between the
two) and learning rates, bit for bit.  The schedule is a piecewise-linear curve through five knots

    step 0 -> initial_lr, end of warmup -> peak_lr, end of stable -> peak_lr,
    end of middle decay -> stable_lr, end of final decay -> final_lr (held afterwards)

and is evaluated as ``a + f * (b - a)`` on the segment the step falls in, which is the same
floating-point expression the reference writes per phase.  The value only reaches the device as the
``lr`` scalar of the next ``FusedAdamW.step()`` launch; no kernel is involved.
"""

from __future__ import annotations

from bisect import bisect_right

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LRScheduler


class WarmupStableDecayLR(LRScheduler):
    """Linear warmup, hold, linear decay, linear decay, hold (src/scheduler.py:11-94)."""

    def __init__(self, optimizer: Optimizer, warmup_steps: int, stable_steps: int, middle_decay_steps: int,
                 final_decay_steps: int, initial_lr: float = 0.0, peak_lr: float | None = None,
                 stable_lr: float | None = None, final_lr: float = 1e-7, last_epoch: int = -1):
        self.warmup_steps = warmup_steps
        self.stable_steps = stable_steps
        self.middle_decay_steps = middle_decay_steps
        self.final_decay_steps = final_decay_steps
        ends = []
        for n in (warmup_steps, stable_steps, middle_decay_steps, final_decay_steps):
            ends.append(n + (ends[-1] if ends else 0))
        # the reference keeps the four phase boundaries under these names (src/scheduler.py:34-39)
        self.phase2_start, self.phase3_start, self.phase4_start, self.phase5_start = ends
        self.initial_lr = initial_lr
        self.peak_lr = optimizer.param_groups[0]["lr"] if peak_lr is None else peak_lr
        self.stable_lr = self.peak_lr * 0.25 if stable_lr is None else stable_lr
        self.final_lr = final_lr
        super().__init__(optimizer, last_epoch)      # takes the first step, which sets lr(0)
        if self.last_epoch == 0:
            for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
                group["lr"] = lr

    def lr_at(self, step: int) -> float:
        """Learning rate of optimizer step number ``step`` (0-based)."""
        ends = (self.phase2_start, self.phase3_start, self.phase4_start, self.phase5_start)
        levels = (self.initial_lr, self.peak_lr, self.peak_lr, self.stable_lr, self.final_lr)
        seg = bisect_right(ends, step)               # zero-length phases are skipped by the bisection
        if seg == 4:
            return self.final_lr
        a, b = levels[seg], levels[seg + 1]
        if seg == 1:
            return b
        begin = ends[seg - 1] if seg else 0
        return a + ((step - begin) / (ends[seg] - begin)) * (b - a)

    def get_lr(self) -> list[float]:
        lr = self.lr_at(self.last_epoch)
        return [lr for _ in self.base_lrs]


def create_scheduler(optimizer: Optimizer, total_steps: int, warmup_steps: int = 2000, initial_lr: float = 0.0,
                     peak_lr: float | None = None, stable_lr: float | None = None,
                     final_lr: float | None = None) -> WarmupStableDecayLR:
    """The paper's split of the steps left after warmup: 52 % at the peak, 13 % for the final decay
    (both rounded down), the rest for the middle decay; ``stable_lr`` defaults to 25 % and
    ``final_lr`` to 2.5 % of the peak (src/scheduler.py:97-159)."""
    warmup_steps = min(warmup_steps, total_steps)
    rest = total_steps - warmup_steps
    stable_steps = int(rest * 0.52)
    final_decay_steps = int(rest * 0.13)
    if peak_lr is None:
        peak_lr = optimizer.param_groups[0]["lr"]
    return WarmupStableDecayLR(
        optimizer, warmup_steps=warmup_steps, stable_steps=stable_steps,
        middle_decay_steps=rest - stable_steps - final_decay_steps, final_decay_steps=final_decay_steps,
        initial_lr=initial_lr, peak_lr=peak_lr,
        stable_lr=peak_lr * 0.25 if stable_lr is None else stable_lr,
        final_lr=peak_lr * 0.025 if final_lr is None else final_lr)
