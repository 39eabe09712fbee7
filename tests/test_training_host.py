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


def test_lr_schedules_match_the_published_lambdas():
    """ap_adapter_amd.training.get_scheduler restates diffusers.optimization.get_scheduler (0.21.2, absent offline; call site
    train_apadapter_v2.py:809-815); the installed transformers ships the same schedule functions (diffusers' file is a copy of it) --
    pinned against their LambdaLR multipliers step by step"""
    import transformers.optimization as TO
    opt = lambda: torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    W, N = 7, 50
    cases = {
        "constant": TO.get_constant_schedule(opt()),
        "constant_with_warmup": TO.get_constant_schedule_with_warmup(opt(), W),
        "linear": TO.get_linear_schedule_with_warmup(opt(), W, N),
        "cosine": TO.get_cosine_schedule_with_warmup(opt(), W, N),
        "cosine_with_restarts": TO.get_cosine_with_hard_restarts_schedule_with_warmup(opt(), W, N, num_cycles=3),
    }
    for name, sched in cases.items():
        f = T.get_scheduler(name, W, N, num_cycles=3)
        lam = sched.lr_lambdas[0]
        for step in range(0, N + 5):
            assert abs(f(step) - lam(step)) < 1e-12, (name, step)
    p_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    lam = TO.get_polynomial_decay_schedule_with_warmup(p_opt, W, N, lr_end=1e-7, power=2.0).lr_lambdas[0]
    f = T.get_scheduler("polynomial", W, N, power=2.0, lr_init=1e-4)
    assert all(abs(f(s_) - lam(s_)) < 1e-12 for s_ in range(N + 5))
    pw = T.get_scheduler("piecewise_constant", step_rules="1:10,0.1:20,0.01")
    assert [pw(0), pw(9), pw(10), pw(19), pw(20), pw(10 ** 6)] == [1.0, 1.0, 0.1, 0.1, 0.01, 0.01]
    import pytest
    with pytest.raises(ValueError):
        T.get_scheduler("linear", W)
    with pytest.raises(ValueError):
        T.get_scheduler("exponential")
    # rules given out of order are walked in sorted step order (diffusers sorts the rule dict); a missing rule string is a clear error
    pw2 = T.get_scheduler("piecewise_constant", step_rules="0.1:20,1:10,0.01")
    assert [pw2(0), pw2(9), pw2(10), pw2(19), pw2(20)] == [1.0, 1.0, 0.1, 0.1, 0.01]
    with pytest.raises(ValueError, match="step_rules"):
        T.get_scheduler("piecewise_constant")
