"""Which lines of the package launch small ATen kernels inside ONE eager training micro-step (cfg 5: batch 4, bf16)?  Same counting as tools/aten_sites.py."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aten_sites import Sites  # noqa: E402


def main():
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_
    dev, dtype, B, La = torch.device("cuda", 0), torch.bfloat16, 4, 32
    with torch.device(dev):
        u = A.AudioLDM2UNet2DConditionModel()
        A.install_ap_adapter(u, None, scale=0.5)
    init_synthetic_(u, 100, on_device=True)
    u = u.to(dev, dtype)
    tr = A.AdapterTrainer(u, lr=1e-4)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(B, 8, 250, 16, generator=g).to(dev)
    ehs = torch.randn(B, 8 + La, 768, generator=g).to(dev)
    ehs1 = torch.randn(B, 16, 1024, generator=g).to(dev)
    m1 = torch.ones(B, 16, device=dev)
    noise = torch.randn(B, 8, 250, 16, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    noisy = A.add_noise(lat, noise, t, tr.alphas_cumprod)
    tr.micro_step(noisy, t, ehs, ehs1, m1, noise)  # warm (caches)
    torch.cuda.synchronize()
    with Sites() as s:
        tr.micro_step(noisy, t, ehs, ehs1, m1, noise)
    torch.cuda.synchronize()
    tot = sum(s.n.values())
    print(f"{tot} ATen calls with device work in one micro-step")
    for (name, site), n in s.n.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 45):
        print(f"{n:6d}  {s.bytes[(name, site)] / n / 1e6:9.3f} MB  {name:42s} {site}")


if __name__ == "__main__":
    main()
