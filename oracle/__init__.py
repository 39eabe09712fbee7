"""CPU oracle for the AP-adapter hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 restatement of the reference's arithmetic for the path named in
BASELINE.json (decoupled cross-attention processor, AudioLDM2 UNet forward, AudioMAE encoder + pooling,
CFG + DDIM loop).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the timed CPU baseline.  The product package
(``ap-adapter_amd``) never imports it and never falls back to it.

Pinning status (see DESIGN.md "Oracle"):
  * attention.py  -- PINNED: bit-compared (fp32, <=1e-6) against the reference's own
    ``APadapter/ap_adapter/attention_processor.py`` imported in the build container; golden vectors
    committed under tests/golden/ (generator: tests/golden/make_golden.py).
  * audiomae.pool -- PINNED against torch.nn.AvgPool2d/MaxPool2d (the ops the reference calls).
  * blocks.py / unet.py / ddim.py / audiomae.vit_block -- PARITY UNPINNED: the arithmetic lives in
    third-party packages that are absent from /root/reference and not installable here
    (diffusers==0.21.2, timm); restated from their published algorithm and anchored on the reference's
    call sites.  The ViT block is additionally cross-checked against transformers' ViTLayer.
"""
