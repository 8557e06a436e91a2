"""Per-step durations of the bench configuration (event per step on the main stream): is the step time stable within a run?"""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4"); os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from mafed_amd import FeatureDistillation, Trainer, VLPythiaConfig, VLPythiaForCausalLM
from mafed_amd.methods import HBMReplayBuffer
dev = torch.device("cuda", 0)
B, P, T = 32, 256, 32
cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)
student = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device=dev, seed=1234)
opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                         distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted", gamma=0.5, distillation_layer=None)
fd._update_model(student)
fd.task_id = 1; fd.num_vision_tokens = P
n_mem = int(os.environ.get("N_MEM", "4000"))
ids = torch.randint(1, cfg.vocab_size, (n_mem, T)); labels = torch.full((n_mem, T), -100); labels[:, -4:] = ids[:, -4:]
feats = torch.randn(n_mem, P, cfg.vision_hidden_size, device=dev).to(torch.bfloat16)
mem = HBMReplayBuffer(B, dev, seed=1)
mem.add({"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels, "patch_embeddings": feats})
fd.mem_dataloader = mem
conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98), weight_decay=0.01, optim="adamw", warmup_perc=0.1)
tr = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=1000, pipeline_optimizer=True)
task = mem.sample()
for i in range(5):
    tr.step(task, i)
torch.cuda.synchronize()
N = 80
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
evs[0].record()
for i in range(N):
    t0 = time.perf_counter()
    tr.step(task, 5 + i)
    host.append((time.perf_counter() - t0) * 1e3)
    evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print("gpu ms per step:", " ".join(f"{x:.1f}" for x in ms))
print("host ms per step:", " ".join(f"{x:.1f}" for x in host))
s = sorted(ms); print(f"min {s[0]:.2f} median {s[N // 2]:.2f} mean {sum(ms) / N:.2f} max {s[-1]:.2f}")
