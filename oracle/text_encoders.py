"""Oracle: prompt encoding (/root/reference/pipeline/pipeline_audioldm2.py:231-270, :272-580).

The three encoders are third-party transformers models in the reference (requirements.txt pins transformers 4.30; the image has a
newer release with the same arithmetic for these models).  PINNED: transformers IS installed in the build / GPU image, so the
checker for them is the dependency itself -- ``ClapTextModelWithProjection`` / ``ClapModel.get_text_features``, ``T5EncoderModel``,
``GPT2Model`` instantiated from tiny seeded configs, their state dicts copied into the HIP modules.  The reference's own glue is
restated here in plain torch: ``add_special_tokens`` and ``AudioLDM2ProjectionModel.forward`` (modeling_audioldm2.py:45-60, :111-145;
diffusers' ModelMixin is not importable, the arithmetic is two nn.Linear and three torch.cat) and ``generate_language_model``
(pipeline_audioldm2.py:231-270) driven through the transformers GPT-2.  TEST INFRASTRUCTURE ONLY.
"""
import torch
import torch.nn.functional as F


def add_special_tokens(hidden_states, attention_mask, sos_token, eos_token):
    B = hidden_states.shape[0]
    if attention_mask is not None:
        ones = attention_mask.new_ones((B, 1))
        attention_mask = torch.cat([ones, attention_mask, ones], dim=-1)
    sos = sos_token.expand(B, 1, -1)
    eos = eos_token.expand(B, 1, -1)
    return torch.cat([sos, hidden_states, eos], dim=1), attention_mask


def projection_model(sd, hidden_states, hidden_states_1, attention_mask, attention_mask_1):
    hs = F.linear(hidden_states, sd["projection.weight"], sd["projection.bias"])
    hs, attention_mask = add_special_tokens(hs, attention_mask, sd["sos_embed"], sd["eos_embed"])
    hs1 = F.linear(hidden_states_1, sd["projection_1.weight"], sd["projection_1.bias"])
    hs1, attention_mask_1 = add_special_tokens(hs1, attention_mask_1, sd["sos_embed_1"], sd["eos_embed_1"])
    return torch.cat([hs, hs1], dim=1), torch.cat([attention_mask, attention_mask_1], dim=-1)


@torch.no_grad()
def generate_language_model(gpt2, inputs_embeds, attention_mask, max_new_tokens):
    """the reference loop with use_cache unset: full re-run per step, mask extended by one (``_update_model_kwargs_for_generation``)"""
    for _ in range(max_new_tokens):
        out = gpt2(inputs_embeds=inputs_embeds, attention_mask=attention_mask, return_dict=True)
        inputs_embeds = torch.cat([inputs_embeds, out.last_hidden_state[:, -1:, :]], dim=1)
        attention_mask = torch.cat([attention_mask, attention_mask.new_ones((attention_mask.shape[0], 1))], dim=-1)
    return inputs_embeds[:, -max_new_tokens:, :]


@torch.no_grad()
def encode_prompt(clap, t5, proj_sd, gpt2, clap_ids, clap_mask, t5_ids, t5_mask, max_new_tokens):
    """encode_prompt for one half of the CFG batch, from token ids (:381-425)"""
    feat = clap.get_text_features(clap_ids, attention_mask=clap_mask)
    feat = getattr(feat, "pooler_output", feat)  # newer transformers return the model output with the normalised feature inside
    pe = feat[:, None, :]
    am = t5_mask.new_ones((clap_ids.shape[0], 1))
    t5h = t5(t5_ids, attention_mask=t5_mask)[0]
    hs, mask = projection_model(proj_sd, pe, t5h, am, t5_mask)
    return t5h, t5_mask, generate_language_model(gpt2, hs, mask, max_new_tokens)
