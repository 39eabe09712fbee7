"""CPU: host logic of the drop-in boundary -- wiring, processor plumbing, checkpoint key scheme, scheduler tables,
C-ABI library loading / symbol export / descriptor layout.  No GPU compute."""
import ctypes as C
import json
import os
import re

import pytest
import torch

import ap_adapter_amd as A
from ap_adapter_amd import _lib as L
from oracle import ddim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INDEX = json.load(open(os.path.join(ROOT, "tests", "golden", "copied_cross_attention_index.json")))["files"]


@pytest.fixture(scope="module")
def unet():
    return A.AudioLDM2UNet2DConditionModel()


def test_geometry_matches_the_shipped_adapter_weights(unet):
    """the 64 copied_cross_attention/*.bin names + shapes pin which attn2 sites take the adapter"""
    assert len(unet.attn_processors) == 256
    names = A.ip_layer_names(unet)
    assert len(names) == 32
    expect = sorted({re.sub(r"_[kv]\.bin$", "", f) for f in INDEX})
    assert sorted(names) == expect
    for n in names:
        attn = unet.get_submodule(n[: -len(".processor")])
        assert list(attn.to_k.weight.shape) == INDEX[n + "_k.bin"]
        assert list(attn.to_v.weight.shape) == INDEX[n + "_v.bin"]


def test_upstream_transformer_slot_layout_selects_the_same_sites():
    """the diffusers AudioLDM2 checkpoints lay the four transformer slots out as (None, 768, None, 1024) (the author's table
    inference.py:16 implies (None, 768, 1024, None); the routing modeling_audioldm2.py:1140-1149 takes either): the adapter sites are
    the 768-wide slot in both, so the same 32 names and shapes"""
    u = A.AudioLDM2UNet2DConditionModel(A.UNetConfig(cross_attention_dim=(None, 768, None, 1024)))
    assert len(u.attn_processors) == 256
    names = A.ip_layer_names(u)
    assert sorted(names) == sorted({re.sub(r"_[kv]\.bin$", "", f) for f in INDEX})
    t5 = [n for n, m in u.named_modules() if n.endswith("attn2") and m.to_k.in_features == 1024]
    assert len(t5) == 32 and all(".attentions." in n and int(n.split(".attentions.")[1].split(".")[0]) % 4 == 3 for n in t5)
    procs = A.install_ap_adapter(u, None, scale=0.5)
    assert sum(hasattr(p, "to_k_ip") for p in procs.values()) == 32


def test_reference_wiring_loop_selects_the_same_sites(unet):
    """inference.py:16-49 verbatim logic: cross[layer_num % 8] over the attn2 processors in attn_processors order"""
    cross = [None, None, 768, 768, 1024, 1024, None, None]
    layer_num, picked = 0, []
    for name in unet.attn_processors.keys():
        if name.endswith("attn1.processor"):
            continue
        if cross[layer_num % 8] == 768:
            picked.append(name)
        layer_num += 1
    assert picked == A.ip_layer_names(unet)


def _tiny_unet():
    return A.AudioLDM2UNet2DConditionModel(A.UNetConfig(block_out_channels=(32, 64, 96, 128), attention_head_dim=4,
                                                        transformer_layers_per_block=1))


def test_do_copy_warm_start_reads_the_reference_file_format(tmp_path):
    """f-1: attention_processor.py:328-344 -- ``do_copy=True`` loads ``copied_cross_attention/<name>_{k,v}.bin`` (pickled fp16
    nn.Parameter, written by copy_weight.py:58-63) as fp32 trainable to_k_ip / to_v_ip.  Through the constructor, through
    install_ap_adapter, and copy_frozen_kv gives the same warm start without files."""
    from make_copied_fixture import write_copied_cross_attention
    u = _tiny_unet()
    files = write_copied_cross_attention(u, str(tmp_path))
    names = A.ip_layer_names(u)
    assert len(files) == 2 * len(names) == 32
    # the constructor path, exactly as the reference calls it
    n0 = names[0]
    C_ = u.get_submodule(n0[: -len(".processor")]).to_q.in_features
    proc = A.IPAttnProcessor2_0(hidden_size=C_, name=n0, cross_attention_dim=768, num_tokens=8, scale=0.5, do_copy=True,
                                copy_dir=str(tmp_path))
    for which, lin in (("k", proc.to_k_ip), ("v", proc.to_v_ip)):
        assert lin.weight.dtype == torch.float32 and lin.weight.requires_grad and isinstance(lin.weight, torch.nn.Parameter)
        assert torch.equal(lin.weight.detach(), files[f"{n0}_{which}.bin"].float())
    # the wiring path: every adapted site
    procs = A.install_ap_adapter(u, None, scale=0.5, do_copy=True, copy_dir=str(tmp_path))
    for n in names:
        assert torch.equal(procs[n].to_k_ip.weight.detach(), files[n + "_k.bin"].float())
        assert torch.equal(procs[n].to_v_ip.weight.detach(), files[n + "_v.bin"].float())
    # the environment-variable default directory
    os.environ["APADAPTER_COPIED_CROSS_ATTENTION"] = str(tmp_path)
    try:
        p2 = A.IPAttnProcessor2_0(hidden_size=C_, name=n0, cross_attention_dim=768, do_copy=True)
        assert torch.equal(p2.to_v_ip.weight.detach(), files[n0 + "_v.bin"].float())
    finally:
        del os.environ["APADAPTER_COPIED_CROSS_ATTENTION"]
    with pytest.raises(FileNotFoundError):
        A.IPAttnProcessor2_0(hidden_size=C_, name="no.such.layer.processor", cross_attention_dim=768, do_copy=True, copy_dir=str(tmp_path))


def test_copy_frozen_kv_equals_what_copy_weight_extracts():
    """copy_weight.py:44-63 saves each adapted layer's frozen attn2.to_k / to_v; copy_frozen_kv installs the same tensors
    directly"""
    from ap_adapter_amd.wiring import copy_frozen_kv
    u = _tiny_unet()
    A.install_ap_adapter(u, None, scale=0.5)
    copy_frozen_kv(u)
    for n, p in u.attn_processors.items():
        if hasattr(p, "to_k_ip"):
            attn = u.get_submodule(n[: -len(".processor")])
            assert torch.equal(p.to_k_ip.weight, attn.to_k.weight) and torch.equal(p.to_v_ip.weight, attn.to_v.weight)
            assert p.to_k_ip.weight.data_ptr() != attn.to_k.weight.data_ptr()  # a copy: training must not move the frozen weight
            assert isinstance(p.to_k_ip.weight, torch.nn.Parameter)


@pytest.mark.skipif(not os.path.isdir("/root/reference/copied_cross_attention"), reason="the reference tree is only in the build container")
def test_do_copy_loads_real_reference_files():
    """build container only: two of the 64 real files (fp16 [C, 768]) through the same loader"""
    u = A.AudioLDM2UNet2DConditionModel()
    d = "/root/reference/copied_cross_attention"
    for n in (A.ip_layer_names(u)[0], A.ip_layer_names(u)[-1]):
        C_ = u.get_submodule(n[: -len(".processor")]).to_q.in_features
        p = A.IPAttnProcessor2_0(hidden_size=C_, name=n, cross_attention_dim=768, num_tokens=8, do_copy=True, copy_dir=d)
        assert list(p.to_k_ip.weight.shape) == INDEX[n + "_k.bin"] == [C_, 768]
        assert p.to_k_ip.weight.dtype == torch.float32 and p.to_k_ip.weight.requires_grad
        assert 0.003 < float(p.to_k_ip.weight.detach().std()) < 0.1  # real weights (std 0.007-0.066)


def test_install_and_checkpoint_roundtrip(tmp_path):
    cfg = A.UNetConfig(block_out_channels=(32, 64, 96, 128), attention_head_dim=4, transformer_layers_per_block=1)
    u = A.AudioLDM2UNet2DConditionModel(cfg)
    procs = A.install_ap_adapter(u, None, scale=0.55, num_tokens=8)
    ip = [p for p in procs.values() if isinstance(p, A.IPAttnProcessor2_0)]
    assert len(procs) == len(u.attn_processors) and len(ip) == 16
    assert all(p.scale == 0.55 and p.num_tokens == 8 and p.cross_attention_dim == 768 for p in ip)
    # processors are registered submodules: their weights show up in state_dict under the reference key scheme
    keys = [k for k in u.state_dict() if ".processor." in k]
    assert len(keys) == 32 and all(k.endswith(("to_k_ip.weight", "to_v_ip.weight")) for k in keys)
    sd = A.adapter_state_dict(u)
    assert sorted(sd) == sorted(keys) and all(v.dtype == torch.float32 for v in sd.values())
    path = tmp_path / "pytorch_model.bin"
    A.save_adapter(u, path)
    sd2 = {k: v * 2 for k, v in A.load_adapter(path).items()}
    u2 = A.AudioLDM2UNet2DConditionModel(cfg)
    A.install_ap_adapter(u2, sd2, scale=0.5)
    for k, v in A.adapter_state_dict(u2).items():
        assert torch.equal(v, sd2[k])


def test_set_attn_processor_contract(unet):
    with pytest.raises(ValueError, match="number of processors"):
        unet.set_attn_processor({"x": A.AttnProcessor2_0()})
    keys = list(unet.attn_processors.keys())
    assert keys[0].startswith("down_blocks") and keys[-1].startswith("mid_block")  # registration order
    d = {k: A.AttnProcessor2_0() for k in keys}
    unet.set_attn_processor(d)
    assert d == {}  # popped, like the reference (modeling_audioldm2.py:567)
    p = A.AttnProcessor2_0()
    unet.set_attn_processor(p)
    assert all(v is p for v in unet.attn_processors.values())


def test_processor_attribute_surface():
    p = A.IPAttnProcessor2_0(hidden_size=256, name="n", cross_attention_dim=768, num_tokens=8, scale=0.5)
    assert hasattr(p, "to_k_ip") and hasattr(p, "to_v_ip")
    assert tuple(p.to_k_ip.weight.shape) == (256, 768) and p.to_k_ip.bias is None
    assert (p.hidden_size, p.cross_attention_dim, p.num_tokens, p.scale, p.name) == (256, 768, 8, 0.5, "n")
    p.to_k_ip.weight = torch.nn.Parameter(torch.zeros(256, 768))  # re-assignment as at inference.py:56-57
    assert isinstance(A.AttnProcessor2_0(hidden_size=1, cross_attention_dim=2), torch.nn.Module)


def test_scheduler_matches_oracle():
    s = A.DDIMScheduler()
    for n in (5, 50, 200):
        s.set_timesteps(n)
        assert torch.equal(s.timesteps, ddim.timesteps(n))
        coef = s.coef_table()
        acp = ddim.alphas_cumprod()
        g = torch.Generator().manual_seed(n)
        x, e = torch.randn(64, generator=g), torch.randn(64, generator=g)
        for i, t in enumerate(s.timesteps.tolist()):
            ref = ddim.ddim_step(e, t, x, n, acp)
            out = coef[i, 0] * x + coef[i, 1] * e
            assert torch.allclose(out, ref, rtol=2e-5, atol=2e-5)
    assert torch.allclose(s.alphas_cumprod, ddim.alphas_cumprod())


def test_pipeline_rejects_out_of_scope_stages(unet):
    pipe = A.AudioLDM2Pipeline(unet)
    with pytest.raises(NotImplementedError, match="text prompts need prompt_encoder"):
        pipe(prompt="jazz", output_type="latent")
    e = torch.zeros(1, 16, 1024)
    with pytest.raises(NotImplementedError, match="AutoencoderKL"):  # the default output_type is the reference's "np" (:775)
        pipe(prompt_embeds=e, negative_prompt_embeds=e, generated_prompt_embeds=e, negative_generated_prompt_embeds=e,
             attention_mask=e[..., 0], negative_attention_mask=e[..., 0])
    with pytest.raises(ValueError, match="required"):
        pipe(prompt_embeds=e, output_type="latent")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "apadapter_hip.h")).read()
    declared = set(re.findall(r"\b(apad_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert os.path.exists(L.LIB_PATH), "libapadapter_hip.so missing: run __graft_entry__.build()"
    h = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        getattr(h, name)  # raises AttributeError when a declared symbol is not exported
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)


def test_descriptor_layout_matches_the_header():
    h = A.lib()
    for desc, echo in ((L.GemmDesc, h.apad_echo_gemm_desc), (L.AttnDesc, h.apad_echo_attn_desc),
                       (L.MlpDesc, h.apad_echo_mlp_desc), (L.AttnBwdDesc, h.apad_echo_attn_bwd_desc),
                       (L.XattnDesc, h.apad_echo_xattn_desc)):
        d = desc()
        names = [f[0] for f in desc._fields_]
        for i, n in enumerate(names):
            setattr(d, n, i + 1)
        out = (C.c_double * 64)()
        n = echo(C.byref(d), out, 64)
        assert n == len(names) and list(out)[:n] == [float(i + 1) for i in range(n)]
    # field order in the header == field order of the ctypes mirror
    header = open(os.path.join(ROOT, "include", "apadapter_hip.h")).read()
    for struct, desc in (("apad_gemm_desc", L.GemmDesc), ("apad_attn_desc", L.AttnDesc), ("apad_mlp_desc", L.MlpDesc),
                         ("apad_attn_bwd_desc", L.AttnBwdDesc), ("apad_xattn_desc", L.XattnDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                fields += [re.sub(r"[\s\*]", "", x).split(" ")[-1] for x in re.sub(r"^(const\s+)?\w+\s*\*?\s*", "", decl).split(",")]
        assert fields == [f[0] for f in desc._fields_], (struct, fields)


def test_rejected_arguments_return_errors_not_aborts():
    h = A.lib()
    d = L.GemmDesc()
    assert h.apad_gemm(C.byref(d), None) != 0
    assert b"apad_gemm" in h.apad_last_error()
    a = L.AttnDesc()
    assert h.apad_attention(C.byref(a), None) != 0
    assert h.apad_gemm(None, None) != 0


def test_no_cpu_fallback():
    from ap_adapter_amd import ops
    x = torch.zeros(4, 16, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(x, torch.zeros(8, 16, dtype=torch.bfloat16))
    # the product package must not import the oracle
    src = os.path.join(ROOT, "ap-adapter_amd")
    for f in os.listdir(src):
        if f.endswith(".py"):
            assert not re.search(r"^\s*(from|import)\s+oracle|import_module\(.oracle", open(os.path.join(src, f)).read(), re.M), f


def test_flop_model_is_consistent_with_the_survey():
    import bench
    for La, expect in ((8, 346.3), (32, 347.7), (128, 353.4), (512, 376.2)):
        mine = bench.unet_flops_per_sample(La) / 1e9
        assert abs(mine - expect) / expect < 0.08  # ours excludes the K/V projections hoisted out of the loop


def test_processors_reject_what_is_not_on_the_path_before_any_kernel():
    """reference :359-363 applies a spatial norm (never configured on the AudioLDM2 path): the drop-in processors refuse it loudly instead of
    computing something else, and so a hidden state that is neither [B, N, C] nor the 4-D [B, C, H, W] of :365-367 (accepted since round 4,
    tests/test_gpu_processors.py: *4d* goldens) -- on the host, before any launch (so this runs without a GPU)."""
    from ap_adapter_amd.unet import Attention
    for proc, cross in ((A.AttnProcessor2_0(), None), (A.IPAttnProcessor2_0(256, "t", cross_attention_dim=768, num_tokens=8), 768)):
        attn = Attention(256, cross, 8, 32)
        attn.set_processor(proc)
        ehs = None if cross is None else torch.zeros(1, 40, 768)
        with pytest.raises(ValueError):
            proc(attn, torch.zeros(1, 256, 4, 4, 2), encoder_hidden_states=ehs)
        attn.spatial_norm = torch.nn.Identity()
        with pytest.raises(NotImplementedError):
            proc(attn, torch.zeros(1, 16, 256), encoder_hidden_states=ehs)


def test_condition_assembly_matches_the_oracle_token_order():
    """a-9 (pipeline_audioldm2.py:934-956): text tokens first, audio after; the unconditional half first; cast to the UNet
    dtype.  The product's assembler against the oracle's on the same tensors (data movement only: bit-equal)."""
    from oracle import audiomae as OA
    g = torch.Generator().manual_seed(5)
    gen = torch.randn(6, 8, 768, generator=g)
    a, u = torch.randn(1, 32, 768, generator=g), torch.randn(1, 32, 768, generator=g)
    out = A.AudioLDM2Pipeline.assemble_condition(gen, a, u, torch.bfloat16)
    ref = OA.assemble_condition(gen, a, u).to(torch.bfloat16)
    assert out.dtype == torch.bfloat16 and out.shape == (6, 40, 768) and out.is_contiguous()
    assert torch.equal(out, ref)


def test_task_presets_equal_the_reference():
    """config.py:2-82 -- pinned by the reference's own get_config outputs (tests/golden/task_presets.json, written by
    tests/golden/make_golden.py::make_task_presets from an import of the reference)"""
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "task_presets.json")))
    assert set(A.config.TASKS) == set(want)
    for t in want:
        assert A.get_config(t) == want[t]
    assert [A.config.audio_tokens(A.get_config(t)) for t in ("timbre_transfer", "style_transfer")] == [128, 32]
    with pytest.raises(KeyError):
        A.get_config("no_such_task")


def test_sharded_job_single_rank_order_padding_and_seeds():
    """cfg 4 driver logic without a GPU (stub encoder / denoiser): clips enumerate (file, prompt) pairs, the last batch is padded
    to the job's one geometry, conditions are laid out unconditional-half-first / text-tokens-first, latents depend on the clip
    only"""
    from ap_adapter_amd import sharded as S
    cfg = A.get_config("style_transfer")
    files = [f"a{i}.wav" for i in range(3)]
    clips = S.list_clips(files, cfg, 7)
    assert [c["audio"] for c in clips] == ["a0.wav", "a1.wav", "a2.wav"] * 2 + ["a0.wav"]
    assert [c["prompt"] for c in clips] == ["Jazz style music"] * 3 + ["Rock style music"] * 3 + ["Pop style music"]
    La, calls = A.config.audio_tokens(cfg), []
    enc = lambda path, tp, fp: (torch.full((La, 768), float(int(path[1]))), torch.zeros(La, 768))

    def den(lat, gen, t5, mask, gs):
        calls.append((tuple(lat.shape), tuple(gen.shape), tuple(t5.shape), gs))
        b = lat.shape[0]
        assert torch.equal(gen[:b, 8:], torch.zeros(b, La, 768))             # unconditional half first, audio tokens after the text
        return lat + gen[b:, 8, 0].reshape(b, 1, 1, 1)                        # + the clip's audio id

    out = S.run_sharded(clips, cfg, enc, den, batch=4, latent_shape=(8, 6, 16))
    assert calls == [((4, 8, 6, 16), (8, 8 + La, 768), (8, 16, 1024), 9.5)] * 2  # 7 clips -> two batches of the SAME geometry
    for c in clips:
        assert torch.equal(out[c["index"]], S.clip_latents(c["index"], (8, 6, 16)) + float(int(c["audio"][1])))
    assert torch.equal(S.gather_clips(out, 7)[5], out[5])
