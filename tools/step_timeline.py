"""Launch time line of one MAFED step WITHOUT an external profiler (the in-library kernel profiler's start/stop events): when does
each kernel class first start and last end inside a step, how busy is the chip.  Run on the GPU box: python tools/step_timeline.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import FeatureDistillation, Trainer, VLPythiaConfig, VLPythiaForCausalLM
from mafed_amd.methods import HBMReplayBuffer
from mafed_amd.profiler import KernelProfile

dev = torch.device("cuda", 0)
B, P, T = 32, 256, 32
cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)
student = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device=dev, seed=1234)
opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                         distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted", gamma=0.5,
                         distillation_layer=None)
fd._update_model(student)
fd.task_id, fd.num_vision_tokens = 1, P
g = torch.Generator().manual_seed(1)
n_mem = 8 * B
ids = torch.randint(1, cfg.vocab_size, (n_mem, T), generator=g)
labels = torch.full((n_mem, T), -100, dtype=torch.int64); labels[:, -4:] = ids[:, -4:]
mem = HBMReplayBuffer(B, dev, seed=3)
mem.add({"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels,
         "patch_embeddings": torch.randn(n_mem, P, cfg.vision_hidden_size, generator=g)})
fd.mem_dataloader = mem
conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98), weight_decay=0.01,
                             optim="adamw", warmup_perc=0.1)
tr = Trainer(student, fd, conf, task_id=1, pipeline_optimizer=True)
tb = mem.sample()
for i in range(8):
    tr.step(tb, i)
torch.cuda.synchronize()
N = 6
with KernelProfile() as kp:
    for i in range(N):
        tr.step(tb, 8 + i)
    torch.cuda.synchronize()
recs, st = kp.records(), kp.starts_ms
# step boundaries: the embedding backward (one per step, the last kernel of the dX chain; the clip norm is now one launch per range)
gn = [i for i, r in enumerate(recs) if r[0] == "embed_concat_bwd"]
print("records", len(recs), "steps", len(gn), "span %.2f ms" % (max(s + r[2] for s, r in zip(st, recs)) - min(st)))
for k in range(2, N - 1):
    t0 = st[gn[k]] + recs[gn[k]][2]          # end of the backward of step k = start of the window
    t1 = st[gn[k + 1]] + recs[gn[k + 1]][2]
    inwin = [(s - t0, r) for s, r in zip(st, recs) if t0 <= s < t1]
    print(f"-- window {k}: {t1 - t0:.2f} ms, {len(inwin)} kernels")
    first = {}
    for s, r in sorted(inwin, key=lambda x: x[0]):
        first.setdefault(r[0], s)
    print("   first start (ms): " + ", ".join(f"{k_}:{v:.2f}" for k_, v in sorted(first.items(), key=lambda kv: kv[1])))
    # the earliest layernorm_fwd with a dual output is the student's or the teacher's first layer: list the first 12 kernels
    print("   first kernels: " + " | ".join(f"{s:.2f} {r[0]}({r[2]*1e3:.0f}us)" for s, r in sorted(inwin, key=lambda x: x[0])[:14]))
    print("   embed_concat_fwd starts (teacher, then student): " + ", ".join(f"{s:.2f}" for s, r in sorted(inwin, key=lambda x: x[0]) if r[0] == "embed_concat_fwd"),
          "| adamw first/last: %.2f / %.2f" % (min([s for s, r in inwin if r[0] == "adamw"] or [0]), max([s + r[2] for s, r in inwin if r[0] == "adamw"] or [0])),
          "| ce_fwd: " + ", ".join(f"{s:.2f}" for s, r in inwin if r[0] == "ce_fwd"))
    # busy histogram
    ev = []
    for s, r in inwin:
        ev += [(s, 1), (s + r[2], -1)]
    ev.sort()
    depth, last, hist = 0, 0.0, {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0.0) + (t - last)
        last, depth = t, depth + d
    print("   overlap depth (ms): " + ", ".join(f"{d}:{v:.2f}" for d, v in sorted(hist.items())))
    # the longest stretches with nothing running (start, length, kernel before -> kernel after)
    iv = sorted((s, s + r[2], r[0]) for s, r in inwin)
    gaps, end, prev = [], iv[0][1], iv[0][2]
    for s, e, name in iv[1:]:
        if s > end:
            gaps.append((s - end, end, prev, name))
        if e > end:
            end, prev = e, name
    gaps.sort(reverse=True)
    print("   idle: %d gaps, %.2f ms; longest: " % (len(gaps), sum(g_[0] for g_ in gaps)) +
          " | ".join(f"{g_[0]*1e3:.0f}us at {g_[1]:.2f} ({g_[2]} -> {g_[3]})" for g_ in gaps[:10]))
