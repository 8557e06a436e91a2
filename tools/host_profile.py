"""Host-side cost of enqueueing training steps (run on the GPU box): cProfile over N MAFED steps of the bench configuration.
python3 tools/host_profile.py [steps] [batch]"""
import cProfile
import os
import pstats
import sys
import time
import types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import FeatureDistillation, Trainer, VLPythiaConfig, VLPythiaForCausalLM
from mafed_amd.methods import HBMReplayBuffer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P, T = 256, 32
dev = torch.device("cuda", 0)
cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)
student = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device=dev, seed=1234)
opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                         distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                         gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
fd._update_model(student)
fd.task_id = 1
fd.num_vision_tokens = P
g = torch.Generator().manual_seed(1)
n_mem = 4 * B
ids = torch.randint(1, cfg.vocab_size, (n_mem, T), generator=g)
labels = torch.full((n_mem, T), -100, dtype=torch.int64)
labels[:, -4:] = ids[:, -4:]
mem = HBMReplayBuffer(B, dev, seed=1236)
mem.add({"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels,
         "patch_embeddings": torch.randn(n_mem, P, cfg.vision_hidden_size, generator=g)})
fd.mem_dataloader = mem
conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98),
                             weight_decay=0.01, optim="adamw", warmup_perc=0.1)
tr = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=1000, pipeline_optimizer=True)
batch = mem.sample()
for i in range(5):
    tr.step(batch, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    tr.step(batch, 5 + i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={B}: host enqueue {t_host / N * 1e3:.2f} ms/step, wall {t_all / N * 1e3:.2f} ms/step", flush=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    tr.step(batch, 5 + N + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
