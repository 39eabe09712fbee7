"""Tiny seeded transformers models (the reference's own dependency for CLAP / T5 / GPT-2) shared by the CPU prompt-encoding tests and
by tests/golden/make_text_golden.py, plus what the GPU tests need WITHOUT importing transformers: the matching configurations of the
HIP modules, the stand-in tokenizer and the loader of the committed fixture (importing transformers on a cold GPU box pages in for
minutes, and the driver's GPU suite has a 20-minute budget)."""
import os

import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_encoders_small.safetensors")
# tiny configurations, as keyword arguments of the HIP modules' config classes (tiny_clap / tiny_t5 / tiny_gpt2 build the same models)
CLAP_CFG = lambda heads: dict(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=128,
                              max_position_embeddings=80, projection_dim=32, pad_token_id=1, layer_norm_eps=1e-12)
T5_CFG = dict(vocab_size=100, d_model=64, d_kv=16, d_ff=96, num_layers=2, num_heads=4, relative_attention_num_buckets=32,
              relative_attention_max_distance=128, layer_norm_epsilon=1e-6)
GPT2_CFG = dict(n_positions=64, n_embd=64, n_layer=2, n_head=4, layer_norm_epsilon=1e-5, vocab_size=100)


def load_text_gold():
    from safetensors.torch import load_file
    return load_file(GOLD)


def ours_from_gold(gold, prefix, kind, dev, heads=4):
    """the HIP module of ``kind`` with the fixture's weights (keys ``<prefix>.sd.<parameter name>``)"""
    import ap_adapter_amd.text_encoders as TE
    if kind == "clap":
        o = TE.ClapTextModelWithProjection(TE.ClapTextConfig(**CLAP_CFG(heads)))
    elif kind == "t5":
        o = TE.T5EncoderModel(TE.T5Config(**T5_CFG))
    elif kind == "gpt2":
        o = TE.GPT2Model(TE.GPT2Config(**GPT2_CFG))
    else:
        o = TE.AudioLDM2ProjectionModel(32, 64, 64)
    sd = {k[len(prefix) + 4:]: v for k, v in gold.items() if k.startswith(prefix + ".sd.")}
    o.load_state_dict(sd, strict=True)
    return o.to(dev)


class Tok:
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


PROMPTS = ["a slow piano melody with soft strings", "drums"]


def tiny_clap(heads=4, hidden=64, seed=0):
    from transformers import ClapAudioConfig, ClapConfig, ClapModel, ClapTextConfig
    torch.manual_seed(seed)
    tc = ClapTextConfig(vocab_size=120, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=128,
                        max_position_embeddings=80, projection_dim=32)
    ac = ClapAudioConfig(patch_embeds_hidden_size=8, depths=[1, 1], num_attention_heads=[1, 1], hidden_size=16, num_mel_bins=16, spec_size=32,
                         patch_size=4, patch_stride=[4, 4], window_size=2, projection_dim=32)
    m = ClapModel(ClapConfig(text_config=tc.to_dict(), audio_config=ac.to_dict(), projection_dim=32)).eval()
    return m, tc


def tiny_t5(seed=1):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    c = T5Config(vocab_size=100, d_model=64, d_kv=16, d_ff=96, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    m = T5EncoderModel(c).eval()
    with torch.no_grad():  # default init leaves the relative bias tiny: make it matter
        m.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.normal_(std=1.0)
    return m, c


def tiny_gpt2(seed=2):
    from transformers import GPT2Config, GPT2Model
    torch.manual_seed(seed)
    c = GPT2Config(vocab_size=100, n_positions=64, n_embd=64, n_layer=2, n_head=4)
    m = GPT2Model(c).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.mul_(3.0)  # default std 0.02 makes attention nearly uniform
    return m, c


def ours_from(tm, tc, kind, dev=None):
    """the HIP module with the transformers module's weights"""
    import ap_adapter_amd.text_encoders as TE
    if kind == "clap":
        o = TE.ClapTextModelWithProjection(TE.ClapTextConfig(vocab_size=tc.vocab_size, hidden_size=tc.hidden_size, num_hidden_layers=tc.num_hidden_layers,
                                                             num_attention_heads=tc.num_attention_heads, intermediate_size=tc.intermediate_size,
                                                             max_position_embeddings=tc.max_position_embeddings, projection_dim=tc.projection_dim,
                                                             pad_token_id=tc.pad_token_id, layer_norm_eps=tc.layer_norm_eps))
    elif kind == "t5":
        o = TE.T5EncoderModel(TE.T5Config(vocab_size=tc.vocab_size, d_model=tc.d_model, d_kv=tc.d_kv, d_ff=tc.d_ff, num_layers=tc.num_layers,
                                          num_heads=tc.num_heads, relative_attention_num_buckets=tc.relative_attention_num_buckets,
                                          relative_attention_max_distance=tc.relative_attention_max_distance, layer_norm_epsilon=tc.layer_norm_epsilon))
    else:
        o = TE.GPT2Model(TE.GPT2Config(n_positions=tc.n_positions, n_embd=tc.n_embd, n_layer=tc.n_layer, n_head=tc.n_head,
                                       layer_norm_epsilon=tc.layer_norm_epsilon, vocab_size=tc.vocab_size))
    sd = tm.state_dict()
    missing, unexpected = o.load_state_dict(sd, strict=False)
    assert not missing, missing  # every parameter of the HIP module exists under the same name in the transformers module
    return (o.to(dev) if dev is not None else o), unexpected


# ---- real layer widths (cvssp/audioldm2: CLAP text tower 768 / 12 heads / 3072 at its 512-token padding, flan-t5-large 1024 / 16 heads /
# d_ff 2816, GPT-2 768 / 12 heads), depth and vocabulary cut so the check stays light.  The weights are NOT stored: both sides build them
# from a seed with seeded_weights_ (the generator copies them into the transformers modules), the fixture holds inputs and outputs only.
GOLD_REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_encoders_real_widths.safetensors")
REAL_CLAP_CFG = dict(vocab_size=2000, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=514, projection_dim=512, pad_token_id=1, layer_norm_eps=1e-12)
REAL_T5_CFG = dict(vocab_size=2000, d_model=1024, d_kv=64, d_ff=2816, num_layers=2, num_heads=16, relative_attention_num_buckets=32,
                   relative_attention_max_distance=128, layer_norm_epsilon=1e-6)
REAL_GPT2_CFG = dict(n_positions=1024, n_embd=768, n_layer=2, n_head=12, layer_norm_epsilon=1e-5, vocab_size=2000)


def seeded_weights_(module, seed):
    """fill every parameter of ``module`` (CPU) from a frozen per-parameter stream, in name order: matrices N(0, 0.05^2) (T5's
    relative-position table N(0, 0.5^2) so that it matters), normalisation gains 1 + 0.1 N(0,1), other vectors N(0, 0.02^2)"""
    with torch.no_grad():
        for i, (name, p) in enumerate(sorted(module.named_parameters(), key=lambda kv: kv[0])):
            g = torch.Generator().manual_seed(seed * 1000 + i)
            r = torch.randn(p.shape, generator=g)
            if p.dim() > 1:
                p.copy_(r * (0.5 if "relative_attention_bias" in name else 0.05))
            elif name.endswith("weight") and any(t in name.lower() for t in ("norm", "ln_")):
                p.copy_(1.0 + 0.1 * r)
            else:
                p.copy_(0.02 * r)
    return module


def real_width_modules():
    """the three HIP modules at the real widths with their seeded weights (CPU)"""
    import ap_adapter_amd.text_encoders as TE
    clap = seeded_weights_(TE.ClapTextModelWithProjection(TE.ClapTextConfig(**REAL_CLAP_CFG)), 31)
    t5 = seeded_weights_(TE.T5EncoderModel(TE.T5Config(**REAL_T5_CFG)), 32)
    gpt = seeded_weights_(TE.GPT2Model(TE.GPT2Config(**REAL_GPT2_CFG)), 33)
    return clap, t5, gpt


def real_width_inputs():
    ids = torch.randint(3, 2000, (2, 512), generator=torch.Generator().manual_seed(21))
    mask = torch.ones_like(ids)
    for b, n in enumerate((11, 40)):  # "max_length" padding: a few tokens, then ~500 pads
        mask[b, n:] = 0
        ids[b, n:] = REAL_CLAP_CFG["pad_token_id"]
    tid = torch.randint(0, 2000, (2, 27), generator=torch.Generator().manual_seed(22))
    tmask = torch.ones_like(tid)
    tmask[1, 19:] = 0
    x = torch.randn(2, 33, 768, generator=torch.Generator().manual_seed(23))
    return ids, mask, tid, tmask, x
