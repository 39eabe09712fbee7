"""Writes a ``copied_cross_attention/`` directory in the reference's exact on-disk format: for every adapted layer two files
``<attn path>.processor_k.bin`` / ``..._v.bin``, each a pickled fp16 ``torch.nn.Parameter`` of shape [C, 768] produced by
``torch.save(processor.to_k_ip.weight, path)`` (/root/reference/copy_weight.py:58-63; read back by
attention_processor.py:328-344).  Used by tests/test_host.py with seeded values; shapes follow the UNet that is passed in
(the real AudioLDM2-large files are [256|384|640, 768], see tests/golden/copied_cross_attention_index.json).

    python tests/golden/make_copied_fixture.py <out_dir>      # small-geometry UNet
"""
import os
import sys

import torch


def write_copied_cross_attention(unet, out_dir, seed=0):
    """one (k, v) pair per adapted attn2 site; returns {file name: fp16 tensor}"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import ap_adapter_amd as A
    os.makedirs(out_dir, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    written = {}
    for name in A.ip_layer_names(unet):
        attn = unet.get_submodule(name[: -len(".processor")])
        for which, lin in (("k", attn.to_k), ("v", attn.to_v)):
            w = (torch.randn(lin.weight.shape, generator=g) * 0.03).half()
            fn = f"{name}_{which}.bin"
            torch.save(torch.nn.Parameter(w), os.path.join(out_dir, fn))  # a pickled Parameter, as the reference writes it
            written[fn] = w
    return written


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import ap_adapter_amd as A
    u = A.AudioLDM2UNet2DConditionModel(A.UNetConfig(block_out_channels=(32, 64, 96, 128), attention_head_dim=4,
                                                    transformer_layers_per_block=1))
    print(len(write_copied_cross_attention(u, sys.argv[1])), "files written to", sys.argv[1])
