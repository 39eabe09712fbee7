"""CPU: the vocoder oracle is PINNED to the installed transformers.SpeechT5HifiGan (the class the reference pipeline holds as
``self.vocoder``), and the host module keeps its parameter names."""
import json
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vocoder as OV

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vocoder_small.safetensors")


def load_gold():
    with safe_open(GOLD, "pt") as f:
        cfg = json.loads(f.metadata()["config"])
        t = {k: f.get_tensor(k) for k in f.keys()}
    sd = {k[2:]: v for k, v in t.items() if k.startswith("w.")}
    return cfg, sd, t["input"], t["output"]


def test_oracle_equals_committed_transformers_output():
    cfg, sd, x, y = load_gold()
    out = OV.hifigan(sd, cfg, x)
    assert out.shape == y.shape == (2, 372)
    assert float((out - y).abs().max()) <= 1e-6 * float(y.abs().max())


def test_oracle_equals_transformers_module_full_config():
    """the AudioLDM2 vocoder geometry (64 mel bins, 1024 channels, rates 5-4-2-2-2), seeded weights, 20 frames"""
    tr = pytest.importorskip("transformers")
    from make_vocoder_golden import seeded_vocoder
    cfg = dict(model_in_dim=64, upsample_initial_channel=1024, upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
               resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1, normalize_before=False)
    m = seeded_vocoder(cfg, seed=3, std=0.02)
    x = torch.randn(1, 20, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = m(x)
        out = OV.hifigan({k: v for k, v in m.state_dict().items()}, cfg, x)
    assert out.shape == ref.shape and float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


def test_host_module_has_the_transformers_parameter_names():
    import ap_adapter_amd as A
    from ap_adapter_amd.vocoder import HifiGanConfig, SpeechT5HifiGan
    cfg, sd, _, _ = load_gold()
    m = SpeechT5HifiGan(HifiGanConfig(model_in_dim=cfg["model_in_dim"], upsample_initial_channel=cfg["upsample_initial_channel"],
                                      upsample_rates=tuple(cfg["upsample_rates"]), upsample_kernel_sizes=tuple(cfg["upsample_kernel_sizes"])))
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert all(m.state_dict()[k].shape == sd[k].shape for k in sd)
    m.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="GPU tensor"):  # no CPU fallback: the HIP extension is the product path
        m(torch.zeros(1, 4, cfg["model_in_dim"]))
