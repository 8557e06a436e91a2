"""Frozen CLIP-ViT-L/14 tower (openai/clip-vit-large-patch14 geometry, random weights) at the step's batch: images/s and the
MFMA rate of the whole pass (run on the GPU box).  python3 tools/vision_bench.py [B] [layers]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd.vision import ClipVisionConfig, ClipVisionTower

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = ClipVisionConfig(hidden_size=1024, num_hidden_layers=L, num_attention_heads=16, intermediate_size=4096, image_size=224, patch_size=14)
tower = ClipVisionTower(cfg, compute_dtype=torch.bfloat16, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
with torch.no_grad():
    for p in tower.parameters():
        p.copy_(torch.randn(p.shape, device="cuda", generator=g) * 0.02)
tower._derived = None
pix = torch.randn(B, 3, 224, 224, device="cuda", generator=g).to(torch.bfloat16)
for _ in range(3):
    tower.patch_features(pix)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tower.patch_features(pix)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
S, h, I, H, D = 257, 1024, 4096, 16, 64
rows = B * S
fl = cfg.layers_run * (2.0 * rows * h * (3 * h + h + 2 * I) + 4.0 * B * H * S * S * D) + 2.0 * B * 256 * h * 588
print(f"CLIP-L/14 tower B={B} layers_run={cfg.layers_run}: {best:.3f} ms  {B / best * 1e3:.0f} images/s  {fl / best / 1e9:.1f} TFLOP/s (unpadded flops)", flush=True)
