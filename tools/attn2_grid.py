"""bench.py's fused_attn2 grid alone (every adapted level x every pooling setting), one JSON object.  usage: python tools/attn2_grid.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.fused_attn2_grid(torch.device("cuda:0"), torch.bfloat16, 64, 0.55), indent=1))
