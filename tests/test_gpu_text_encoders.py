"""-m gpu: prompt encoding on the HIP path (fp32 precision mode) against the installed transformers modules -- the reference's own
dependency for CLAP / T5 / GPT-2 -- with their weights copied in, and against oracle/text_encoders.py for the reference's glue."""
import math

import pytest
import torch
import torch.nn.functional as F

from text_models import ours_from, tiny_clap, tiny_gpt2, tiny_t5
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


@pytest.mark.parametrize("heads", [4, 2])  # head dim 16 -> GEMM / softmax / GEMM chain; 32 -> apad_attention with the mask as a key bias
def test_clap_text_features_vs_transformers(dev, heads):
    tm, tc = tiny_clap(heads=heads)
    ours, _ = ours_from(tm, tc, "clap", dev)
    ids = torch.randint(2, tc.vocab_size, (3, 13), generator=torch.Generator().manual_seed(9))
    mask = torch.ones_like(ids)
    mask[1, 9:] = 0
    ids[1, 9:] = tc.pad_token_id
    mask[2, 4:] = 0
    ids[2, 4:] = tc.pad_token_id
    with torch.no_grad():
        ref = tm.get_text_features(ids, attention_mask=mask)
    ref = getattr(ref, "pooler_output", ref)
    out = ours.get_text_features(ids.to(dev), attention_mask=mask.to(dev))
    assert out.shape == ref.shape == (3, tc.projection_dim)
    assert rel_err(out, ref) < TOL


def test_t5_encoder_vs_transformers(dev):
    tm, tc = tiny_t5()
    ours, _ = ours_from(tm, tc, "t5", dev)
    ids = torch.randint(0, tc.vocab_size, (2, 21), generator=torch.Generator().manual_seed(10))
    mask = torch.ones_like(ids)
    mask[1, 15:] = 0
    with torch.no_grad():
        ref = tm(ids, attention_mask=mask)[0]
    out = ours(ids.to(dev), attention_mask=mask.to(dev))[0]
    assert out.shape == ref.shape
    valid = mask.bool()
    assert rel_err(out.cpu()[valid], ref[valid]) < TOL
    # the relative-position bias is live in this comparison: zeroing it must change the result
    with torch.no_grad():
        ours.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.zero_()
    assert rel_err(ours(ids.to(dev), attention_mask=mask.to(dev))[0].cpu()[valid], ref[valid]) > 1e-3


def test_gpt2_inputs_embeds_vs_transformers(dev):
    tm, tc = tiny_gpt2()
    ours, _ = ours_from(tm, tc, "gpt2", dev)
    x = R(2, 11, tc.n_embd, seed=11)
    mask = torch.ones(2, 11, dtype=torch.long)
    mask[1, 3:6] = 0
    with torch.no_grad():
        ref = tm(inputs_embeds=x, attention_mask=mask).last_hidden_state
    out = ours(x.to(dev), attention_mask=mask.to(dev))
    valid = mask.bool()
    assert rel_err(out.cpu()[valid], ref[valid]) < TOL
    # causal: the first positions do not see what follows
    x2 = x.clone()
    x2[:, 8:] += 1.0
    out2 = ours(x2.to(dev), attention_mask=mask.to(dev))
    assert torch.equal(out2[:, :8], out[:, :8])


def test_encode_prompt_vs_reference_chain(dev):
    """encode_prompt for one CFG half from token ids (pipeline_audioldm2.py:381-425): CLAP feature as one token, T5 states, projection with
    SOS / EOS, 8 generated GPT-2 vectors -- against transformers + the restated glue"""
    import ap_adapter_amd.text_encoders as TE
    from oracle import text_encoders as O
    clap_t, cc = tiny_clap(heads=2)
    t5_t, c5 = tiny_t5()
    gpt_t, cg = tiny_gpt2()
    torch.manual_seed(12)
    proj = TE.AudioLDM2ProjectionModel(cc.projection_dim, c5.d_model, cg.n_embd)
    with torch.no_grad():
        for p in proj.parameters():
            p.copy_(torch.randn(p.shape) * (0.2 if p.dim() > 1 else 0.5))
    psd = {k: v.detach().clone() for k, v in proj.state_dict().items()}
    enc = TE.PromptEncoder(ours_from(clap_t, cc, "clap", dev)[0], ours_from(t5_t, c5, "t5", dev)[0], proj.to(dev), ours_from(gpt_t, cg, "gpt2", dev)[0])
    g = torch.Generator().manual_seed(13)
    B = 2
    cid = torch.randint(2, cc.vocab_size, (B, 16), generator=g)
    cm = torch.ones_like(cid)
    cm[1, 7:] = 0
    cid[1, 7:] = cc.pad_token_id
    tid = torch.randint(0, c5.vocab_size, (B, 9), generator=g)
    tmask = torch.ones_like(tid)
    tmask[0, 6:] = 0
    ref_t5, ref_mask, ref_gen = O.encode_prompt(clap_t, t5_t, psd, gpt_t, cid, cm, tid, tmask, 8)
    t5h, m, gen = enc.encode(cid.to(dev), cm.to(dev), tid.to(dev), tmask.to(dev), max_new_tokens=8)
    assert gen.shape == ref_gen.shape == (B, 8, cg.n_embd) and torch.equal(m.cpu(), ref_mask)
    assert rel_err(t5h.cpu()[tmask.bool()], ref_t5[tmask.bool()]) < TOL
    assert rel_err(gen, ref_gen) < 1e-4  # 8 auto-regressive passes


class _Tok:
    """a stand-in tokenizer with the transformers call signature (the real ones need vocabulary files): whitespace words hashed into the
    vocabulary, `cls` / `eos` framing, padding to max_length or to the longest"""

    def __init__(self, vocab, pad_id, model_max_length, bos=None, eos=None):
        self.vocab, self.pad_id, self.model_max_length, self.bos, self.eos = vocab, pad_id, model_max_length, bos, eos

    def __call__(self, texts, padding=True, max_length=None, truncation=True, return_tensors="pt"):
        from types import SimpleNamespace
        rows = []
        for t in texts:
            ids = [3 + (sum(map(ord, w)) % (self.vocab - 3)) for w in t.split()]
            ids = ([self.bos] if self.bos is not None else []) + ids + ([self.eos] if self.eos is not None else [])
            rows.append(ids[: max_length or self.model_max_length])
        L = (max_length or self.model_max_length) if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_id, dtype=torch.long)
        mask = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
            mask[i, : len(r)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)


def test_pipeline_encode_prompt_from_text(dev):
    """AudioLDM2Pipeline.encode_prompt(prompt=[...]) (:272-580): CLAP ids padded to model_max_length, T5 ids to the longest prompt, negative
    prompts "" padded to the positive T5 length, per-waveform repeat, [negative; positive] stacking -- against the transformers chain on the
    same token ids"""
    import ap_adapter_amd as A
    import ap_adapter_amd.text_encoders as TE
    from oracle import text_encoders as O
    clap_t, cc = tiny_clap(heads=2)
    t5_t, c5 = tiny_t5()
    gpt_t, cg = tiny_gpt2()
    torch.manual_seed(14)
    proj = TE.AudioLDM2ProjectionModel(cc.projection_dim, c5.d_model, cg.n_embd)
    with torch.no_grad():
        for p in proj.parameters():
            p.copy_(torch.randn(p.shape) * (0.2 if p.dim() > 1 else 0.5))
    psd = {k: v.detach().clone() for k, v in proj.state_dict().items()}
    enc = A.PromptEncoder(ours_from(clap_t, cc, "clap", dev)[0], ours_from(t5_t, c5, "t5", dev)[0], proj.to(dev), ours_from(gpt_t, cg, "gpt2", dev)[0])
    tok1 = _Tok(cc.vocab_size, cc.pad_token_id, 24, bos=0, eos=2)
    tok2 = _Tok(c5.vocab_size, 0, 32, eos=1)
    pipe = A.AudioLDM2Pipeline(None, prompt_encoder=enc, tokenizer=tok1, tokenizer_2=tok2)
    prompts = ["a slow piano melody with soft strings", "drums"]
    pe, am, ge = pipe.encode_prompt(prompts, dev, 2, True, max_new_tokens=8)
    # reference chain on the same ids
    c_pos, t_pos = tok1(prompts, padding="max_length", max_length=24), tok2(prompts, padding=True, max_length=32)
    r_t5, r_m, r_gen = O.encode_prompt(clap_t, t5_t, psd, gpt_t, c_pos.input_ids, c_pos.attention_mask, t_pos.input_ids, t_pos.attention_mask, 8)
    Lt = r_t5.shape[1]
    c_neg, t_neg = tok1(["", ""], padding="max_length", max_length=24), tok2(["", ""], padding="max_length", max_length=Lt)
    n_t5, n_m, n_gen = O.encode_prompt(clap_t, t5_t, psd, gpt_t, c_neg.input_ids, c_neg.attention_mask, t_neg.input_ids, t_neg.attention_mask, 8)
    rep = lambda t: t.repeat_interleave(2, dim=0)
    ref_pe, ref_am, ref_ge = torch.cat([rep(n_t5), rep(r_t5)]), torch.cat([rep(n_m), rep(r_m)]), torch.cat([rep(n_gen), rep(r_gen)])
    assert pe.shape == ref_pe.shape == (8, Lt, c5.d_model) and ge.shape == (8, 8, cg.n_embd)
    assert torch.equal(am.cpu(), ref_am)
    valid = ref_am.bool()
    assert rel_err(pe.cpu()[valid], ref_pe[valid]) < TOL and rel_err(ge, ref_ge) < 1e-4
    # no encoder -> the text entry point says what is missing
    with pytest.raises(NotImplementedError, match="prompt_encoder"):
        A.AudioLDM2Pipeline(None).encode_prompt(prompts, dev, 1, True)


def test_real_widths_vs_transformers(dev):
    """the cvssp/audioldm2 widths (CLAP text tower 768 / 12 heads / 3072 at its 512-token padding, flan-t5-large 1024 / 16 heads / d_ff 2816,
    GPT-2 768 / 12 heads), depth and vocabulary cut so the test stays light -- the per-layer arithmetic and envelopes are the real ones"""
    from transformers import ClapAudioConfig, ClapConfig, ClapModel, ClapTextConfig, GPT2Config, GPT2Model, T5Config, T5EncoderModel
    torch.manual_seed(20)
    tc = ClapTextConfig(vocab_size=2000, num_hidden_layers=2)  # 768 / 12 / 3072 / 514 positions / projection 512
    ac = ClapAudioConfig(patch_embeds_hidden_size=8, depths=[1, 1], num_attention_heads=[1, 1], hidden_size=16, num_mel_bins=16, spec_size=32,
                         patch_size=4, patch_stride=[4, 4], window_size=2, projection_dim=512)
    clap = ClapModel(ClapConfig(text_config=tc.to_dict(), audio_config=ac.to_dict(), projection_dim=512)).eval()
    ours, _ = ours_from(clap, tc, "clap", dev)
    ids = torch.randint(3, 2000, (2, 512), generator=torch.Generator().manual_seed(21))
    mask = torch.ones_like(ids)
    for b, n in enumerate((11, 40)):  # "max_length" padding: a few tokens, then 500 pads
        mask[b, n:] = 0
        ids[b, n:] = tc.pad_token_id
    with torch.no_grad():
        ref = clap.get_text_features(ids, attention_mask=mask)
    ref = getattr(ref, "pooler_output", ref)
    assert rel_err(ours.get_text_features(ids.to(dev), attention_mask=mask.to(dev)), ref) < TOL

    c5 = T5Config(vocab_size=2000, d_model=1024, d_kv=64, d_ff=2816, num_layers=2, num_heads=16, feed_forward_proj="gated-gelu")
    t5 = T5EncoderModel(c5).eval()
    with torch.no_grad():
        t5.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.normal_(std=0.5)
        for p in t5.parameters():
            if p.dim() > 1 and p.shape[0] != 32:
                p.mul_(0.05)  # T5's init (std 1 on the embedding) saturates a random-weight model
    o5, _ = ours_from(t5, c5, "t5", dev)
    tid = torch.randint(0, 2000, (2, 27), generator=torch.Generator().manual_seed(22))
    tm_ = torch.ones_like(tid)
    tm_[1, 19:] = 0
    with torch.no_grad():
        r5 = t5(tid, attention_mask=tm_)[0]
    assert rel_err(o5(tid.to(dev), attention_mask=tm_.to(dev))[0].cpu()[tm_.bool()], r5[tm_.bool()]) < TOL

    cg = GPT2Config(vocab_size=2000, n_layer=2)  # 768 / 12 heads / 1024 positions
    gp = GPT2Model(cg).eval()
    og, _ = ours_from(gp, cg, "gpt2", dev)
    x = R(2, 33, 768, seed=23)
    with torch.no_grad():
        rg = gp(inputs_embeds=x).last_hidden_state
    assert rel_err(og(x.to(dev)), rg) < TOL
