"""CPU: pin the optimizer / noising restatements of oracle/train.py to torch's own implementations (the reference calls
torch.optim.AdamW at train_apadapter_v2.py:763-769 and clip_grad_norm_ at :975), and run the oracle's adapter gradient
on a tiny geometry."""
import torch

from oracle import train as OT


def test_adamw_restatement_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    p = torch.nn.Parameter(torch.randn(257, 33, generator=g) * 0.1)
    opt = torch.optim.AdamW([p], lr=3e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    q, m, v = p.detach().clone(), torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(257, 33, generator=g) * 0.3
        p.grad = grad.clone()
        opt.step()
        q, m, v = OT.adamw_update(q, grad, m, v, step, 3e-3)
        assert torch.allclose(q, p.detach(), rtol=1e-5, atol=1e-7)


def test_clip_restatement_matches_torch():
    g = torch.Generator().manual_seed(1)
    for scale in (0.01, 5.0):
        ps = [torch.nn.Parameter(torch.zeros(50, 7)), torch.nn.Parameter(torch.zeros(13))]
        grads = [torch.randn(50, 7, generator=g) * scale, torch.randn(13, generator=g) * scale]
        for p_, g_ in zip(ps, grads):
            p_.grad = g_.clone()
        total = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        coef, tot = OT.clip_coef(grads, 1.0)
        assert torch.allclose(tot, total, rtol=1e-6)
        for p_, g_ in zip(ps, grads):
            assert torch.allclose(p_.grad, g_ * coef, rtol=1e-6, atol=1e-9)


def test_add_noise_is_the_ddpm_forward_process():
    acp = OT.alphas_cumprod()
    assert acp.shape == (1000,) and 0.99 < float(acp[0]) < 1 and float(acp[-1]) < 0.01
    x, n = torch.ones(2, 8, 4, 4), torch.full((2, 8, 4, 4), 2.0)
    t = torch.tensor([0, 999])
    y = OT.add_noise(x, n, t)
    assert torch.allclose(y[0], acp[0].sqrt() + 2 * (1 - acp[0]).sqrt())
    assert torch.allclose(y[1], acp[999].sqrt() + 2 * (1 - acp[999]).sqrt())
