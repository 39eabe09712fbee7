"""CPU: prompt-encoding host logic -- parameter naming against the installed transformers modules (the reference's dependency), the T5
bucket function against transformers' own, the reference glue restated in oracle/text_encoders.py, loud failure without a GPU."""
import pytest
import torch

import ap_adapter_amd.text_encoders as TE
from oracle import text_encoders as O
from text_models import ours_from, tiny_clap, tiny_gpt2, tiny_t5


def test_parameter_names_and_shapes_are_the_transformers_ones():
    for kind, (tm, tc) in (("clap", tiny_clap()), ("t5", tiny_t5()), ("gpt2", tiny_gpt2())):
        o, unexpected = ours_from(tm, tc, kind)
        sd = tm.state_dict()
        for k, v in o.state_dict().items():
            assert k in sd and tuple(sd[k].shape) == tuple(v.shape), (kind, k)
            assert torch.equal(sd[k], v)
        # what is left over is not on the text path: CLAP's audio tower / logit scales, index buffers
        for k in unexpected:
            assert kind == "clap" and (k.startswith(("audio_model.", "audio_projection.", "logit_scale")) or k.endswith(("position_ids", "token_type_ids"))), k


def test_t5_bucket_function_equals_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention
    pos = torch.arange(300)
    rel = pos[None, :] - pos[:, None]
    for nb, md in ((32, 128), (16, 64)):
        assert torch.equal(TE.t5_relative_position_bucket(rel, nb, md), T5Attention._relative_position_bucket(rel, True, nb, md))


def test_projection_glue_order_and_masks():
    """modeling_audioldm2.py:111-145: [sos, clap, eos, sos_1, t5..., eos_1], masks extended by ones either side"""
    torch.manual_seed(0)
    sd = {"projection.weight": torch.randn(8, 4), "projection.bias": torch.randn(8), "projection_1.weight": torch.randn(8, 6),
          "projection_1.bias": torch.randn(8), "sos_embed": torch.full((8,), 1.0), "eos_embed": torch.full((8,), 2.0),
          "sos_embed_1": torch.full((8,), 3.0), "eos_embed_1": torch.full((8,), 4.0)}
    a, b = torch.randn(2, 1, 4), torch.randn(2, 5, 6)
    m1 = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]])
    hs, mask = O.projection_model(sd, a, b, torch.ones(2, 1, dtype=torch.long), m1)
    assert hs.shape == (2, 3 + 7, 8) and mask.tolist() == [[1, 1, 1, 1, 1, 1, 1, 0, 0, 1], [1] * 10]
    assert float(hs[0, 0, 0]) == 1.0 and float(hs[0, 2, 0]) == 2.0 and float(hs[0, 3, 0]) == 3.0 and float(hs[0, 9, 0]) == 4.0
    assert torch.allclose(hs[:, 1], torch.nn.functional.linear(a[:, 0], sd["projection.weight"], sd["projection.bias"]))
    # the host-side twin used by the HIP module
    b8 = torch.randn(2, 5, 8)
    h2, m2 = TE.add_special_tokens(b8, m1, sd["sos_embed_1"], sd["eos_embed_1"])
    assert torch.equal(h2, O.add_special_tokens(b8, m1, sd["sos_embed_1"], sd["eos_embed_1"])[0]) and m2.shape == (2, 7)


def test_generation_loop_against_transformers_gpt2():
    """the hidden-state auto-regression (pipeline_audioldm2.py:231-270): 8 new vectors, each the last hidden state of the previous run"""
    tm, tc = tiny_gpt2()
    x = torch.randn(2, 5, tc.n_embd, generator=torch.Generator().manual_seed(3))
    mask = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 1, 0, 1]])
    g = O.generate_language_model(tm, x, mask, 4)
    assert g.shape == (2, 4, tc.n_embd)
    first = tm(inputs_embeds=x, attention_mask=mask).last_hidden_state[:, -1]
    assert torch.allclose(g[:, 0], first, atol=1e-6)


def test_no_cpu_path():
    tm, tc = tiny_t5()
    o, _ = ours_from(tm, tc, "t5")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        o(torch.zeros(1, 8, dtype=torch.long))
    g, _ = ours_from(*tiny_gpt2(), "gpt2")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 8, 64))


def test_committed_fixture_equals_the_installed_transformers_outputs():
    """tests/golden/text_encoders_small.safetensors (what the GPU tests compare the HIP path with) re-derived here from the live transformers
    modules: weights bit-equal, outputs to 1e-6 (summation order of the CPU BLAS may differ between machines)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_text_golden as G
    from text_models import load_text_gold
    fresh, gold = G.build(), load_text_gold()
    assert set(fresh) == set(gold)
    for k, v in fresh.items():
        if ".sd." in k or v.dtype != torch.float32:
            assert torch.equal(v, gold[k]), k
        else:
            assert float((v - gold[k]).abs().max()) <= 1e-5 * max(float(gold[k].abs().max()), 1.0), k


def test_real_width_fixture_equals_the_installed_transformers_outputs():
    """tests/golden/text_encoders_real_widths.safetensors re-derived from the live transformers modules carrying the seeded weights (the
    GPU test builds the same weights from the same seed and never imports transformers)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_text_golden as G
    from safetensors.torch import load_file
    from text_models import GOLD_REAL
    fresh, gold = G.build_real_widths(), load_file(GOLD_REAL)
    assert set(fresh) == set(gold)
    for k, v in fresh.items():
        assert float((v - gold[k]).abs().max()) <= 1e-5 * max(float(gold[k].abs().max()), 1.0), k
