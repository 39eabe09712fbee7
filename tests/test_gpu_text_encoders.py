"""-m gpu: prompt encoding on the HIP path (fp32 precision mode) against the outputs of the installed transformers modules -- the
reference's own dependency for CLAP / T5 / GPT-2 -- and of the reference's glue around them (oracle/text_encoders.py), read from the
committed fixture tests/golden/text_encoders_small.safetensors (weights, inputs, outputs; tests/golden/make_text_golden.py wrote it,
tests/test_oracle_text_encoders.py re-derives it from transformers on the CPU).  transformers itself is NOT imported here: on a cold GPU
box its import alone pages in for minutes (measured: 70-320 s inside the suite), and the driver's GPU suite has a 20-minute budget.
The comparison at the real layer widths reads tests/golden/text_encoders_real_widths.safetensors (outputs only; weights from a seed)."""
import math

import pytest
import torch
import torch.nn.functional as F

import os

from text_models import CLAP_CFG, GPT2_CFG, PROMPTS, T5_CFG, Tok, load_text_gold, ours_from_gold
from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


def R(*shape, seed=0, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


def test_rmsnorm_l2norm_embedding_kernels(dev):
    from ap_adapter_amd import ops
    x, g = R(37, 96, seed=1) * 3, 1 + 0.1 * R(96, seed=2)
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * g
    assert rel_err(ops.rms_norm(x.to(dev), g.to(dev), 1e-6), ref) < 1e-6
    assert rel_err(ops.l2_normalize(x.to(dev)), F.normalize(x, dim=-1)) < 1e-6
    xb = x.bfloat16()
    assert rel_err(ops.rms_norm(xb.to(dev), g.bfloat16().to(dev), 1e-6), ref) < 2e-2
    table = R(50, 24, seed=3)
    ids = torch.tensor([[0, 49, 7], [7, 3, 100]])
    out = ops.embedding(table.to(dev), ids.to(dev))
    assert torch.equal(out[0].cpu(), table[ids[0]]) and torch.equal(out[1, :2].cpu(), table[ids[1, :2]])
    assert float(out[1, 2].abs().max()) == 0  # an id outside the table reads as zeros
    with pytest.raises(RuntimeError, match="int64 GPU"):
        ops.embedding(table.to(dev), ids.int().to(dev))


def test_softmax_rows_with_bias_and_masked_rows(dev):
    from ap_adapter_amd import ops
    x = R(40, 24, seed=4) * 2
    b = R(40, 24, seed=5)
    b[:, 20:] = float("-inf")
    b[3] = float("-inf")  # a fully masked row: zeros, not NaN
    ref = torch.softmax(x * 0.5 + b, dim=-1)
    ref[3] = 0
    out = ops.softmax_rows(x.to(dev), 0.5, bias=b.to(dev))
    assert torch.isfinite(out).all() and rel_err(out, ref) < 1e-6
    assert float(out[:, 20:].abs().max()) == 0


@pytest.mark.parametrize("act", ["relu", "gelu_tanh", "geglu_tanh", "tanh"])
def test_fp32_mode_epilogues(dev, act):
    from ap_adapter_amd import ops
    M, K, N = 70, 64, 96
    rows = 2 * N if act == "geglu_tanh" else N
    x, w, b = R(M, K, seed=6), R(rows, K, seed=7, std=0.2), R(rows, seed=8, std=0.5)
    y = F.linear(x, w, b)
    ref = {"relu": lambda: F.relu(y), "gelu_tanh": lambda: F.gelu(y, approximate="tanh"), "tanh": lambda: torch.tanh(y),
           "geglu_tanh": lambda: y[:, :N] * F.gelu(y[:, N:], approximate="tanh")}[act]()
    assert rel_err(ops.linear(x.to(dev), w.to(dev), b.to(dev), act=act), ref) < 1e-5
    if act != "tanh":
        with pytest.raises(RuntimeError, match="fp32-mode epilogue"):
            ops.linear(x.bfloat16().to(dev), w.bfloat16().to(dev), b.bfloat16().to(dev), act=act)


@pytest.fixture(scope="module")
def gold():
    return load_text_gold()


@pytest.mark.parametrize("heads", [4, 2])  # head dim 16 -> GEMM / softmax / GEMM chain; 32 -> apad_attention with the mask as a key bias
def test_clap_text_features_vs_transformers(dev, gold, heads):
    pre = f"clap{heads}"
    ours = ours_from_gold(gold, pre, "clap", dev, heads=heads)
    ids, mask, ref = gold[pre + ".ids"], gold[pre + ".mask"], gold[pre + ".out"]
    out = ours.get_text_features(ids.to(dev), attention_mask=mask.to(dev))
    assert out.shape == ref.shape == (3, 32)
    assert rel_err(out, ref) < TOL


def test_t5_encoder_vs_transformers(dev, gold):
    ours = ours_from_gold(gold, "t5", "t5", dev)
    ids, mask, ref = gold["t5.ids"], gold["t5.mask"], gold["t5.out"]
    out = ours(ids.to(dev), attention_mask=mask.to(dev))[0]
    assert out.shape == ref.shape
    valid = mask.bool()
    assert rel_err(out.cpu()[valid], ref[valid]) < TOL
    # the relative-position bias is live in this comparison: zeroing it must change the result
    with torch.no_grad():
        ours.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.zero_()
    assert rel_err(ours(ids.to(dev), attention_mask=mask.to(dev))[0].cpu()[valid], ref[valid]) > 1e-3


def test_gpt2_inputs_embeds_vs_transformers(dev, gold):
    ours = ours_from_gold(gold, "gpt2", "gpt2", dev)
    x, mask, ref = gold["gpt2.x"], gold["gpt2.mask"], gold["gpt2.out"]
    out = ours(x.to(dev), attention_mask=mask.to(dev))
    valid = mask.bool()
    assert rel_err(out.cpu()[valid], ref[valid]) < TOL
    # causal: the first positions do not see what follows
    x2 = x.clone()
    x2[:, 8:] += 1.0
    out2 = ours(x2.to(dev), attention_mask=mask.to(dev))
    assert torch.equal(out2[:, :8], out[:, :8])


def _encoder(gold, dev):
    import ap_adapter_amd as A
    return A.PromptEncoder(ours_from_gold(gold, "clap2", "clap", dev, heads=2), ours_from_gold(gold, "t5", "t5", dev),
                           ours_from_gold(gold, "proj", "proj", dev), ours_from_gold(gold, "gpt2", "gpt2", dev))


def test_encode_prompt_vs_reference_chain(dev, gold):
    """encode_prompt for one CFG half from token ids (pipeline_audioldm2.py:381-425): CLAP feature as one token, T5 states, projection with
    SOS / EOS, 8 generated GPT-2 vectors -- against transformers + the restated glue (fixture)"""
    enc = _encoder(gold, dev)
    D = lambda k: gold[k].to(dev)
    t5h, m, gen = enc.encode(D("enc.cid"), D("enc.cm"), D("enc.tid"), D("enc.tmask"), max_new_tokens=8)
    ref_t5, ref_mask, ref_gen = gold["enc.t5"], gold["enc.mask"], gold["enc.gen"]
    assert gen.shape == ref_gen.shape == (2, 8, GPT2_CFG["n_embd"]) and torch.equal(m.cpu(), ref_mask)
    valid = gold["enc.tmask"].bool()
    assert rel_err(t5h.cpu()[valid], ref_t5[valid]) < TOL
    assert rel_err(gen, ref_gen) < 1e-4  # 8 auto-regressive passes


def test_pipeline_encode_prompt_from_text(dev, gold):
    """AudioLDM2Pipeline.encode_prompt(prompt=[...]) (:272-580): CLAP ids padded to model_max_length, T5 ids to the longest prompt, negative
    prompts "" padded to the positive T5 length, per-waveform repeat, [negative; positive] stacking -- against the transformers chain on the
    same token ids (fixture)"""
    import ap_adapter_amd as A
    enc = _encoder(gold, dev)
    tok1 = Tok(CLAP_CFG(2)["vocab_size"], CLAP_CFG(2)["pad_token_id"], 24, bos=0, eos=2)
    tok2 = Tok(T5_CFG["vocab_size"], 0, 32, eos=1)
    pipe = A.AudioLDM2Pipeline(None, prompt_encoder=enc, tokenizer=tok1, tokenizer_2=tok2)
    pe, am, ge = pipe.encode_prompt(PROMPTS, dev, 2, True, max_new_tokens=8)
    ref_pe, ref_am, ref_ge = gold["pipe.pe"], gold["pipe.am"], gold["pipe.ge"]
    assert pe.shape == ref_pe.shape and ge.shape == ref_ge.shape == (8, 8, GPT2_CFG["n_embd"])
    assert torch.equal(am.cpu(), ref_am)
    valid = ref_am.bool()
    assert rel_err(pe.cpu()[valid], ref_pe[valid]) < TOL and rel_err(ge, ref_ge) < 1e-4
    # no encoder -> the text entry point says what is missing
    with pytest.raises(NotImplementedError, match="prompt_encoder"):
        A.AudioLDM2Pipeline(None).encode_prompt(PROMPTS, dev, 1, True)


def test_real_widths_vs_transformers_fixture(dev):
    """the cvssp/audioldm2 widths (CLAP text tower 768 / 12 heads / 3072 at its 512-token padding, flan-t5-large 1024 / 16 heads / d_ff 2816,
    GPT-2 768 / 12 heads), depth and vocabulary cut so the test stays light -- the per-layer arithmetic and envelopes are the real ones.
    Weights come from a seed on both sides (text_models.seeded_weights_); the expected outputs are those of the installed transformers
    modules carrying the same weights (tests/golden/make_text_golden.py::build_real_widths), so transformers is not imported here."""
    from safetensors.torch import load_file
    from text_models import GOLD_REAL, real_width_inputs, real_width_modules
    gold = load_file(GOLD_REAL)
    clap, t5, gpt = real_width_modules()
    ids, mask, tid, tmask, x = real_width_inputs()
    # fp32 on both sides; only the summation order differs (reductions of 768 ... 3072 terms, two layers deep): 5e-5 of max
    tol = 5e-5
    assert rel_err(clap.to(dev).get_text_features(ids.to(dev), attention_mask=mask.to(dev)), gold["clap.out"]) < tol
    assert rel_err(t5.to(dev)(tid.to(dev), attention_mask=tmask.to(dev))[0].cpu()[tmask.bool()], gold["t5.out"][tmask.bool()]) < tol
    assert rel_err(gpt.to(dev)(x.to(dev)), gold["gpt2.out"]) < tol
