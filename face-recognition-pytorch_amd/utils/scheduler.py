"""Per-epoch learning-rate schedule with the reference's name and constructor
(/root/reference/utils/scheduler.py:5-87): linear warm-up from min_lr to max_lr over `warmup_steps`, then a
half cosine back to min_lr over the rest of the cycle; cycles restart, each `cycle_mult` times longer and with
max_lr scaled by `gamma`.  Restated from that behaviour as a pure function of the epoch counter."""
import math


def _cycle_position(epoch, first, warm, mult):
    """-> (cycle index, step inside the cycle, cycle length) for a 0-based epoch counter."""
    cyc, length, start = 0, first, 0
    while epoch >= start + length:
        start += length
        cyc += 1
        length = int((length - warm) * mult) + warm
    return cyc, epoch - start, length


class CosineAnnealingWarmupRestarts:
    def __init__(self, optimizer, first_cycle_steps, cycle_mult=1.0, max_lr=0.1, min_lr=0.001, warmup_steps=0,
                 gamma=1.0, last_epoch=-1):
        assert warmup_steps < first_cycle_steps
        self.optimizer = optimizer
        self.first_cycle_steps, self.cycle_mult = first_cycle_steps, cycle_mult
        self.base_max_lr, self.max_lr, self.min_lr = max_lr, max_lr, min_lr
        self.warmup_steps, self.gamma = warmup_steps, gamma
        self.last_epoch = last_epoch
        self.cycle, self.step_in_cycle, self.cur_cycle_steps = 0, last_epoch, first_cycle_steps
        self.step()                                   # torch's _LRScheduler.__init__ performs one step() ...
        for group in optimizer.param_groups:          # ... and the reference then resets every group to min_lr
            group["lr"] = min_lr                      # (init_lr(), :42-46): epoch 0 always trains at min_lr

    def lr_at(self, epoch):
        cyc, pos, length = _cycle_position(epoch, self.first_cycle_steps, self.warmup_steps, self.cycle_mult)
        peak = self.base_max_lr * (self.gamma ** cyc)
        if pos < self.warmup_steps:
            return self.min_lr + (peak - self.min_lr) * pos / self.warmup_steps
        frac = (pos - self.warmup_steps) / (length - self.warmup_steps)
        return self.min_lr + (peak - self.min_lr) * (1.0 + math.cos(math.pi * frac)) / 2.0

    def get_last_lr(self):
        return [self.lr_at(max(self.last_epoch, 0)) for _ in self.optimizer.param_groups]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else int(math.floor(epoch))
        self.cycle, self.step_in_cycle, self.cur_cycle_steps = _cycle_position(
            self.last_epoch, self.first_cycle_steps, self.warmup_steps, self.cycle_mult)
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        lr = self.lr_at(self.last_epoch)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
