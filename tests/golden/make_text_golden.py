"""tests/golden/text_encoders_small.safetensors: seeded weights, inputs and the outputs of the installed transformers modules
(ClapModel.get_text_features, T5EncoderModel, GPT2Model) for the tiny configurations of tests/text_models.py, plus the reference chain
(oracle/text_encoders.py glue around those modules) for encode_prompt -- so that the GPU tests never import transformers.  Run in the build
container:  python tests/golden/make_text_golden.py      (tests/test_oracle_text_encoders.py re-derives every output from transformers and
fails when the fixture is stale)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    import ap_adapter_amd.text_encoders as TE
    from oracle import text_encoders as O
    from text_models import PROMPTS, Tok, ours_from, tiny_clap, tiny_gpt2, tiny_t5
    out = {}

    def put_sd(prefix, tm, tc, kind):
        o, _ = ours_from(tm, tc, kind)
        for k, v in o.state_dict().items():
            out[f"{prefix}.sd.{k}"] = v.detach().clone().contiguous()

    clap = {}
    for heads in (4, 2):
        tm, tc = tiny_clap(heads=heads)
        clap[heads] = (tm, tc)
        ids = torch.randint(2, tc.vocab_size, (3, 13), generator=torch.Generator().manual_seed(9))
        mask = torch.ones_like(ids)
        mask[1, 9:] = 0
        ids[1, 9:] = tc.pad_token_id
        mask[2, 4:] = 0
        ids[2, 4:] = tc.pad_token_id
        with torch.no_grad():
            ref = tm.get_text_features(ids, attention_mask=mask)
        ref = getattr(ref, "pooler_output", ref)
        put_sd(f"clap{heads}", tm, tc, "clap")
        out[f"clap{heads}.ids"], out[f"clap{heads}.mask"], out[f"clap{heads}.out"] = ids, mask, ref.contiguous()

    t5, c5 = tiny_t5()
    ids = torch.randint(0, c5.vocab_size, (2, 21), generator=torch.Generator().manual_seed(10))
    mask = torch.ones_like(ids)
    mask[1, 15:] = 0
    with torch.no_grad():
        ref = t5(ids, attention_mask=mask)[0]
    put_sd("t5", t5, c5, "t5")
    out["t5.ids"], out["t5.mask"], out["t5.out"] = ids, mask, ref.contiguous()

    gpt, cg = tiny_gpt2()
    x = torch.randn(2, 11, cg.n_embd, generator=torch.Generator().manual_seed(11))
    mask = torch.ones(2, 11, dtype=torch.long)
    mask[1, 3:6] = 0
    with torch.no_grad():
        ref = gpt(inputs_embeds=x, attention_mask=mask).last_hidden_state
    put_sd("gpt2", gpt, cg, "gpt2")
    out["gpt2.x"], out["gpt2.mask"], out["gpt2.out"] = x, mask, ref.contiguous()

    # encode_prompt chain (CLAP with 2 heads -> head dim 32, the apad_attention route)
    clap_t, cc = clap[2]
    torch.manual_seed(12)
    proj = TE.AudioLDM2ProjectionModel(cc.projection_dim, c5.d_model, cg.n_embd)
    with torch.no_grad():
        for p in proj.parameters():
            p.copy_(torch.randn(p.shape) * (0.2 if p.dim() > 1 else 0.5))
    psd = {k: v.detach().clone() for k, v in proj.state_dict().items()}
    for k, v in psd.items():
        out[f"proj.sd.{k}"] = v.contiguous()
    g = torch.Generator().manual_seed(13)
    cid = torch.randint(2, cc.vocab_size, (2, 16), generator=g)
    cm = torch.ones_like(cid)
    cm[1, 7:] = 0
    cid[1, 7:] = cc.pad_token_id
    tid = torch.randint(0, c5.vocab_size, (2, 9), generator=g)
    tmask = torch.ones_like(tid)
    tmask[0, 6:] = 0
    r_t5, r_m, r_gen = O.encode_prompt(clap_t, t5, psd, gpt, cid, cm, tid, tmask, 8)
    out.update({"enc.cid": cid, "enc.cm": cm, "enc.tid": tid, "enc.tmask": tmask, "enc.t5": r_t5.contiguous(), "enc.mask": r_m.contiguous(),
                "enc.gen": r_gen.contiguous()})

    # pipeline.encode_prompt from text through the stand-in tokenizers: positive prompts, "" negatives padded to the positive T5 length,
    # two waveforms per prompt, [negative; positive]
    tok1, tok2 = Tok(cc.vocab_size, cc.pad_token_id, 24, bos=0, eos=2), Tok(c5.vocab_size, 0, 32, eos=1)
    c_pos, t_pos = tok1(PROMPTS, padding="max_length", max_length=24), tok2(PROMPTS, padding=True, max_length=32)
    p_t5, p_m, p_gen = O.encode_prompt(clap_t, t5, psd, gpt, c_pos.input_ids, c_pos.attention_mask, t_pos.input_ids, t_pos.attention_mask, 8)
    Lt = p_t5.shape[1]
    c_neg, t_neg = tok1(["", ""], padding="max_length", max_length=24), tok2(["", ""], padding="max_length", max_length=Lt)
    n_t5, n_m, n_gen = O.encode_prompt(clap_t, t5, psd, gpt, c_neg.input_ids, c_neg.attention_mask, t_neg.input_ids, t_neg.attention_mask, 8)
    rep = lambda t: t.repeat_interleave(2, dim=0)
    out["pipe.pe"] = torch.cat([rep(n_t5), rep(p_t5)]).contiguous()
    out["pipe.am"] = torch.cat([rep(n_m), rep(p_m)]).contiguous()
    out["pipe.ge"] = torch.cat([rep(n_gen), rep(p_gen)]).contiguous()
    return out


def build_real_widths():
    """outputs of the installed transformers modules at the real layer widths, on the seeded weights of text_models.real_width_modules
    (copied INTO the transformers modules: the HIP modules use their parameter names)"""
    from text_models import REAL_CLAP_CFG, REAL_GPT2_CFG, REAL_T5_CFG, real_width_inputs, real_width_modules
    from transformers import ClapAudioConfig, ClapConfig, ClapModel, ClapTextConfig, GPT2Config, GPT2Model, T5Config, T5EncoderModel
    ours_clap, ours_t5, ours_gpt = real_width_modules()

    def fill(hf, ours):
        sd = ours.state_dict()
        missing, unexpected = hf.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected  # every HIP-module parameter exists under the same name in the transformers module
        return [k for k in missing]

    tc = ClapTextConfig(**{k: v for k, v in REAL_CLAP_CFG.items()})
    ac = ClapAudioConfig(patch_embeds_hidden_size=8, depths=[1, 1], num_attention_heads=[1, 1], hidden_size=16, num_mel_bins=16, spec_size=32,
                         patch_size=4, patch_stride=[4, 4], window_size=2, projection_dim=512)
    clap = ClapModel(ClapConfig(text_config=tc.to_dict(), audio_config=ac.to_dict(), projection_dim=512)).eval()
    left = [k for k in fill(clap, ours_clap) if k.startswith("text_") and "position_ids" not in k and "token_type_ids" not in k]
    assert not left, left  # the whole text branch was overwritten (the audio tower keeps its own init: unused by get_text_features)
    c5 = T5Config(feed_forward_proj="gated-gelu", **{k: v for k, v in REAL_T5_CFG.items()})
    t5 = T5EncoderModel(c5).eval()
    assert not fill(t5, ours_t5)
    cg = GPT2Config(**REAL_GPT2_CFG)
    gpt = GPT2Model(cg).eval()
    left = [k for k in fill(gpt, ours_gpt) if not k.endswith(".attn.bias") and not k.endswith(".attn.masked_bias")]
    assert not left, left
    ids, mask, tid, tmask, x = real_width_inputs()
    with torch.no_grad():
        rc = clap.get_text_features(ids, attention_mask=mask)
        rc = getattr(rc, "pooler_output", rc)
        r5 = t5(tid, attention_mask=tmask)[0]
        rg = gpt(inputs_embeds=x).last_hidden_state
    return {"clap.out": rc.contiguous(), "t5.out": r5.contiguous(), "gpt2.out": rg.contiguous()}


if __name__ == "__main__":
    from safetensors.torch import save_file
    from text_models import GOLD, GOLD_REAL
    real = build_real_widths()
    save_file(real, GOLD_REAL)
    print(f"wrote {GOLD_REAL}: {len(real)} tensors, {sum(v.numel() * v.element_size() for v in real.values()) / 1e6:.2f} MB")
    o = {k: v.detach().clone().contiguous() for k, v in build().items()}  # (some entries alias one another: masks handed through)
    save_file(o, GOLD)
    print(f"wrote {GOLD}: {len(o)} tensors, {sum(v.numel() * v.element_size() for v in o.values()) / 1e6:.2f} MB")
