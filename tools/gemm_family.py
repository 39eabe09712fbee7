"""Per-shape table of the conv / plain-GEMM family of ONE denoise step at the bench geometry (batch 32, La 32): every apad_gemm
descriptor of an eager step is recorded, identical shapes are merged, and each shape is re-launched from a copy of its descriptor inside a
hipGraph (20 launches x 3 replays, HIP events) -> launches per step, us per launch, TFLOP/s, fraction of the 2.5 PF dense peak.
usage: python tools/gemm_family.py [min_us]   (APAD_LIB_PATH selects a variant library)"""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ap_adapter_amd as A
from ap_adapter_amd import _lib as L, ops
from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs
from bench import time_kernel_graphed

dev = torch.device("cuda:0")
dtype = torch.bfloat16
B, La = int(os.environ.get("B", "32")), 32
with torch.device(dev):
    unet = A.AudioLDM2UNet2DConditionModel()
    A.install_ap_adapter(unet, None, scale=0.55)
init_synthetic_(unet, 100, on_device=True)
unet = unet.to(dev, dtype)
inp = synthetic_inputs(B, La, seed=0)
pipe = A.AudioLDM2Pipeline(unet)
ge = pipe.assemble_condition(inp["generated_prompt_embeds"].to(dev), inp["audio_tokens"].to(dev), inp["uncond_audio_tokens"].to(dev), dtype)
pe, am = inp["prompt_embeds"].to(dev, dtype), inp["attention_mask"].to(dev)
H, W, Cc = 250, 16, 8
sched = pipe.scheduler
sched.set_timesteps(200)
coef = sched.coef_table().to(dev)
step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
lat = inp["latents"].to(dev).float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
unet_in = lat.to(dtype)
unet.set_kv_cache(True)
unet.precompute_time_tables(sched.timesteps.to(dev), step_ptr)


def step():
    eps2 = unet.forward_nhwc(unet_in, H, W, None, ge, pe, None, am, batch_repeat=2)
    ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, 9.5)
    ops.step_advance(step_ptr)


h = L.lib()
seen = collections.OrderedDict()  # key -> [count, descriptor bytes]
AM = {0: "plain", 1: "conv3x3", 2: "patch16", 4: "conv1d"}


class Proxy:
    def __getattr__(self, name):
        f = getattr(h, name)
        if name != "apad_gemm":
            return f

        def g(dref, stream):
            d = dref._obj
            key = (AM.get(d.a_mode, d.a_mode), d.M, d.K, d.N, d.Cin, d.stride, d.Hup, d.epilogue, d.out_mode, bool(d.residual), bool(d.a2),
                   bool(d.rowstat_in), bool(d.rowstat_out), bool(d.rowgroup_bias), bool(d.w_halo))
            if key not in seen:
                seen[key] = [0, bytes(C.string_at(C.addressof(d), C.sizeof(d)))]
            seen[key][0] += 1
            return f(dref, stream)
        return g


with torch.no_grad():
    step()
    torch.cuda.synchronize()
    real = L._lib
    L._lib = Proxy()
    step()
    torch.cuda.synchronize()
    L._lib = real

min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
print(f"# apad_gemm launches of one denoise step (batch {B}, La {La}); isolated, hipGraph-timed (warm L2 / Infinity Cache); peak 2500 TF/s")
print("# per_step  a_mode    M       K      N     Cin  stride up  epi out res a2  lnfold stat_out tab halo |   us     TF/s   frac  | step_us")
tot = 0.0
rows = []
# the step's own operands are gone by now: every pointer of a recorded descriptor is re-aimed at scratch of sufficient size
_scr = {}


def scratch(name, nbytes, fill=None):
    t = _scr.get(name)
    if t is None or t.numel() * 2 < nbytes:
        n = (int(nbytes) + 255) // 2
        t = (torch.randn(n, device=dev) * (0.02 if name == "w" else 0.5)).to(dtype) if fill is None else torch.zeros(n, device=dev, dtype=dtype)
        _scr[name] = t
    return t.data_ptr()


for key, (cnt, raw) in seen.items():
    d = L.GemmDesc.from_buffer_copy(raw)
    rows_a = d.M if d.a_mode == 0 else (d.M // max(1, d.Hout * d.Wout)) * d.Hin * d.Win
    d.a = scratch("a", rows_a * max(d.lda, d.Cin, 1) * 2 + 4096)
    d.w = scratch("w", d.N * (2 if d.epilogue in (3, 7) else 1) * d.ldw * 2 + 4096)
    d.out = scratch("out", d.M * max(d.ldo, d.N) * 2 * 3 + 4096)
    if d.out2: d.out2 = scratch("out2", d.M * max(d.ldo, d.N) * 2 + 4096)
    if d.out3: d.out3 = scratch("out3", d.M * max(d.ldo, d.N) * 4 + 4096)
    if d.out4: d.out4 = scratch("out4", d.M * max(d.ldo, d.N) * 2 + 4096)
    if d.bias: d.bias = scratch("bias", d.N * 8 + 4096)
    if d.residual: d.residual = scratch("res", d.M * max(d.ldr, d.N) * 2 + 4096)
    if d.rowgroup_bias:
        d.rowgroup_bias = scratch("rg", (d.M // max(1, d.rows_per_group) + 256) * max(d.ld_rg, d.N) * 2 + 4096, fill=0)
    if d.step_ptr: d.step_ptr = step_ptr.zero_().data_ptr()
    if d.rowstat_out: d.rowstat_out = scratch("rso", d.M * (d.N // 64 + 1) * 8 + 4096, fill=0)
    if d.rowstat_in: d.rowstat_in = scratch("rsi", d.M * (d.rowstat_in_tiles + 1) * 8 + 4096, fill=0)
    if d.ln_colsum: d.ln_colsum = scratch("lnc", d.N * 16 + 4096, fill=0)
    if d.ln_bias: d.ln_bias = scratch("lnb", d.N * 16 + 4096, fill=0)
    if d.a2: d.a2 = scratch("a2", d.M * max(d.lda2, 1) * 2 + 4096)
    if d.workspace: d.workspace = scratch("ws", d.workspace_bytes + 4096, fill=0)  # (w_halo stays: the packed weights are cached by ops)
    fn = lambda: L.check(h.apad_gemm(C.byref(d), ops._stream()), "apad_gemm")
    ms = time_kernel_graphed(fn)
    am_, M, K, N, Cin, stride, Hup, epi, om, res, a2, lnf, so, tab, halo = key
    fl = 2.0 * M * K * N * (2 if epi in (3, 7) else 1)
    tf = fl / (ms * 1e-3) / 1e12
    rows.append((cnt * ms * 1e3, cnt, key, ms * 1e3, tf))
    tot += cnt * ms * 1e3
for su, cnt, key, us, tf in sorted(rows, key=lambda r: -r[0]):
    if us < min_us:
        continue
    am_, M, K, N, Cin, stride, Hup, epi, om, res, a2, lnf, so, tab, halo = key
    print(f"  {cnt:5d}     {am_:8s} {M:7d} {K:6d} {N:5d} {Cin:5d}   {stride}    {int(Hup > 0)}   {epi}   {om}   {int(res)}   {int(a2)}    {int(lnf)}      {int(so)}      {int(tab)}   {int(halo)}  | {us:7.1f} {tf:7.1f}  {tf / 2500:5.3f} | {su:8.1f}")
print(f"# total {tot / 1e3:.3f} ms per step over {sum(r[1] for r in rows)} launches")
