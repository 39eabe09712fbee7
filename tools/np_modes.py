"""guided noise_pred of the first DDIM step (bench geometry) in the three precision modes: pairwise differences.  usage: python tools/np_modes.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import ap_adapter_amd as A
import bench
from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = argparse.Namespace(batch=B, guidance=9.5, la=32)
with torch.device(dev):
    unet = A.AudioLDM2UNet2DConditionModel()
    A.install_ap_adapter(unet, None, scale=0.55)
init_synthetic_(unet, 100, on_device=True)
unet = unet.to(dev, torch.bfloat16)
inp = synthetic_inputs(B, 32, seed=0)
if os.environ.get("NP_STREAMS"):
    unet.low_res_streams = tuple(torch.cuda.Stream() for _ in range(int(os.environ["NP_STREAMS"])))
    unet.low_res_levels = 1
out = {}
for name, dt, graph in (("bf16", torch.bfloat16, True), ("f16", torch.float16, True), ("f32", torch.float32, False)):
    r, npred = bench.precision_leg(A, unet, inp, args, dev, dt, steps=2, graph=graph)
    out[name] = npred
    print(name, r["ms_per_step"], "max|np|", float(npred.abs().max()), "mean|np|", float(npred.abs().mean()), flush=True)
for a, b in (("bf16", "f16"), ("bf16", "f32"), ("f16", "f32")):
    d = out[a] - out[b]
    cos = float((out[a] * out[b]).sum() / (out[a].norm() * out[b].norm()))
    print(a, "vs", b, "max-abs", float(d.abs().max()), "mean-abs", float(d.abs().mean()), "cos", cos)
