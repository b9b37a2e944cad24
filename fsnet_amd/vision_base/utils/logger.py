"""Running-mean loss logger (reference vision_base/utils/logger.py:6-75).  One difference, on
purpose: update() keeps device tensors and only synchronises when values are read (log()/avg), so
the per-key `.item()` host syncs of the reference (logger.py:50) do not serialise the GPU timeline."""


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self._sum, self.count, self._pending = 0.0, 0, []

    def update(self, val, n=1):
        self._pending.append((val, n))
        self.count += n

    def _flush(self):
        for val, n in self._pending:
            self._sum += float(val) * n
        self._pending = []

    @property
    def sum(self):
        self._flush()
        return self._sum

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


class LossLogger(object):
    def __init__(self, recorder=None, data_split="train"):
        self.recorder, self.data_split = recorder, data_split
        self.reset()

    def reset(self):
        self.loss_stats, self.hm = {}, {}

    def update(self, loss_dict):
        for k, v in loss_dict.items():
            self.loss_stats.setdefault(k, AverageMeter()).update(v.detach() if hasattr(v, "detach") else v)

    def update_hm(self, hm):
        self.hm = hm

    def log(self, step):
        if self.recorder is None:
            return
        for k, m in self.loss_stats.items():
            self.recorder.add_scalar("%s/%s" % (k, self.data_split), m.avg, step)
