"""CPU: the data side of the trainer (reference train_apadapter_v2.py:443-454 condition dropout, :990-1006 checkpoint
rotation) -- host logic only, no kernels."""
import os

import torch

from ap_adapter_amd import training as T


class _Seq:
    def __init__(self, vals):
        self.vals = list(vals)

    def random(self):
        return self.vals.pop(0)


def test_condition_dropout_thresholds():
    mels = [torch.ones(4, 4) * (i + 1) for i in range(6)]
    texts = [f"t{i}" for i in range(6)]
    t, m = T.apply_condition_dropout(texts, mels, _Seq([0.049, 0.05, 0.099, 0.1, 0.149, 0.15]))
    assert t == ["", "t1", "t2", "", "", "t5"]
    zeroed = [bool((x == 0).all()) for x in m]
    assert zeroed == [False, True, True, True, True, False]
    assert texts[0] == "t0" and bool((mels[1] != 0).all())          # inputs untouched


def test_dropout_rates_are_five_percent_each():
    import random
    rng = random.Random(0)
    n, c = 40000, [0, 0, 0]
    for _ in range(n // 8):
        t, m = T.apply_condition_dropout(["x"] * 8, [torch.ones(1)] * 8, rng)
        for ti, mi in zip(t, m):
            dt, dm = ti == "", bool((mi == 0).all())
            c[0] += dt and not dm
            c[1] += dm and not dt
            c[2] += dt and dm
    assert all(abs(k / n - 0.05) < 0.006 for k in c)


def test_checkpoint_rotation(tmp_path):
    out = str(tmp_path)
    for s in (100, 200, 300, 1000):
        os.makedirs(os.path.join(out, f"checkpoint-{s}"))
    assert T.rotate_checkpoints(out, None) == []
    removed = T.rotate_checkpoints(out, 3)                            # room for the one about to be written
    assert removed == ["checkpoint-100", "checkpoint-200"]
    assert sorted(os.listdir(out), key=lambda d: int(d.split("-")[1])) == ["checkpoint-300", "checkpoint-1000"]
    assert T.rotate_checkpoints(out, 5) == []
