"""tests/golden/vocoder_small.safetensors: inputs, seeded weights and the output of ``transformers.SpeechT5HifiGan`` itself (the
class the reference pipeline holds as ``self.vocoder``, pipeline_audioldm2.py:583-590) on a small configuration.
    python tests/golden/make_vocoder_golden.py
"""
import json
import os

import torch
from safetensors.torch import save_file
from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig

SMALL = dict(model_in_dim=16, upsample_initial_channel=64, upsample_rates=[5, 2], upsample_kernel_sizes=[16, 4],
             resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1, normalize_before=False)


def seeded_vocoder(cfg_dict, seed=0, std=0.05):
    torch.manual_seed(seed)
    m = SpeechT5HifiGan(SpeechT5HifiGanConfig(**cfg_dict)).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * std)
    return m


if __name__ == "__main__":
    m = seeded_vocoder(SMALL)
    x = torch.randn(2, 37, SMALL["model_in_dim"], generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = m(x)
    t = {"input": x, "output": y}
    t.update({"w." + k: v.detach().clone() for k, v in m.state_dict().items()})
    here = os.path.dirname(os.path.abspath(__file__))
    save_file(t, os.path.join(here, "vocoder_small.safetensors"), metadata={"config": json.dumps(SMALL), "source": "transformers.SpeechT5HifiGan"})
    print("wrote vocoder_small.safetensors", y.shape, float(y.abs().max()))
