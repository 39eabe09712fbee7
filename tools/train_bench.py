"""Training-step timing at BASELINE config 5 (AudioLDM2-large geometry, B = 4 per GPU, bf16 compute, fp32 master,
random t per sample, La = 32): forward + loss + backward + clip + AdamW.  usage: python tools/train_bench.py [steps] [B] [La]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ap_adapter_amd as A
from ap_adapter_amd.synthetic import init_synthetic_

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
La = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev, dtype = torch.device("cuda:0"), torch.bfloat16
u = A.AudioLDM2UNet2DConditionModel()
A.install_ap_adapter(u, None, scale=0.5)
init_synthetic_(u, 100)
u = u.to(dev, dtype)
tr = A.AdapterTrainer(u, lr=1e-4)
g = torch.Generator().manual_seed(0)
lat = torch.randn(B, 8, 250, 16, generator=g).to(dev)
ehs = torch.randn(B, 8 + La, 768, generator=g).to(dev)
ehs1 = torch.randn(B, 16, 1024, generator=g).to(dev)
m1 = torch.ones(B, 16, device=dev)
graphed = os.environ.get("APAD_TRAIN_GRAPH", "1") != "0"
replay = tr.capture_micro_step(B, 250, 16, 8 + La, 16) if graphed else None
losses = []
torch.cuda.reset_peak_memory_stats()
for i in range(steps + 2):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    noise = torch.randn(B, 8, 250, 16, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    if graphed:
        noisy = A.add_noise(lat, noise, t, tr.alphas_cumprod)
        losses.append(replay(noisy, t, ehs, ehs1, m1, noise).clone())
        tr.optimizer_step()
    else:
        losses.append(tr.train_step(lat, noise, t, ehs, ehs1, m1))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"train_step_ms": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 2), "B": B, "La": La, "steps": steps, "hipgraph": graphed,
                  "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "finite": bool(torch.isfinite(torch.stack(losses)).all()),
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "adapter_params": int(tr.master.numel())}))
