"""Headline benchmark: train samples/s of the MAFED distill step, VLPythia-410M, 256 image + 32 text tokens,
batch 32 per GPU (BASELINE.json metric / configs[2]); N = 1, 2, 4, 8 GPUs of one node, one process per GPU over RCCL.

A "step" = one full MAFED optimiser step on one synthetic memory batch resident in HBM: student forward with hidden
states + frozen-teacher forward + replay cross-entropy + (L-1)-layer per-modality distillation MSE + backward +
global-norm clip + AdamW (accumulate 1, replay on every step: SURVEY.md section 8d "primary").  Nothing is skipped or
cached inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 410m] [--batch 32] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

``python bench.py --gpus N`` without a launcher starts the N ranks itself (a torch.distributed.run child, before anything touches
the GPU) and fails if the node has fewer than N GPUs or fewer than N ranks join -- it never reports a 1-GPU number for N > 1.

The JSON line carries, beside the contract's keys: ``roofline`` (dominant kernel = the bf16 MFMA GEMM: algorithmic flops / the
kernel's own execution time inside the step, measured live with start/stop events on the launches -- the quantity rocprofv3
--kernel-trace reports, see profiles/), ``kernels`` (the same for attention and for every HBM-bound kernel against 8 TB/s),
``secondary`` (the reference's blended schedule: bs 16 x accumulate 4, 3 plain-CE + 1 MAFED micro-batch per optimiser step,
scripts/run_seed42.sh:51-70) and ``cpu_baseline``.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Two HIP runtime options are read when the runtime initialises, hence before torch is imported: they are this benchmark's settings
# (flags below, an exported variable wins over the flag's default) and the JSON line records what was in force.
#   HIP_FORCE_DEV_KERNARG=1  kernel arguments in device memory: shaves the per-launch dependency latency of the ~600 launches of a step
#                            (33.93 -> 33.76 ms in a same-box A/B); --no-dev-kernarg leaves the runtime's default
#   GPU_MAX_HW_QUEUES=4      the step's streams are scheduled for four hardware queues (main | teacher + a dW stream | two dW streams |
#                            optimiser + loader): 3 queues -> 36.6 ms, 4 -> 34.8, 5 -> 42.3, 8 -> 41.0 in one box; --hw-queues N overrides
def _early_flag(name, default=None, is_bool=False):
    for i, a in enumerate(sys.argv[1:], 1):
        if a == name:
            return True if is_bool else (sys.argv[i + 1] if i + 1 < len(sys.argv) else default)
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return False if is_bool else default


_hwq = _early_flag("--hw-queues")
if _hwq is not None:
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(_hwq))
else:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
if not _early_flag("--no-dev-kernarg", is_bool=True):
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
_rccl_ch = _early_flag("--rccl-channels")
if _rccl_ch is not None:  # RCCL reads these when the communicator is created
    os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(int(_rccl_ch))

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md chip table (never the 2:1-sparsity figure)


def algorithmic_flops_per_sample(h, L, V, P, T, dv, distill=True):
    """SURVEY.md section 8d conventions: 2 flops/MAC; 24 h^2 per token per layer; causal attention 4 h S(S+1)/2 per layer;
    LM head on the T text positions; projector 2 (dv h + h^2) P; backward = 2x forward; teacher = (L-2)/L of the layer
    stack + projector, no head, no backward."""
    S = P + T
    stack = L * (24.0 * h * h * S + 4.0 * h * S * (S + 1) / 2.0)
    head = 2.0 * h * V * T
    proj = 2.0 * (dv * h + h * h) * P
    fwd = stack + head + proj
    total = 3.0 * fwd
    if distill:
        total += stack * (L - 2) / L + proj
    return total


def lm_head_rows(B, T, hint, bf16, sparse):
    """Rows the LM head runs on for a batch of B samples: B*T dense, or B x (rows per sample of the row-sparse head, model.py) when the
    batch carries the loader's ``max_label_rows`` hint."""
    if not sparse or hint is None:
        return B * T
    rc = next((r for r in range(max(2, hint + 1), T + 1) if (B * r) % 128 == 0), None) if bf16 else max(2, hint + 1)
    return B * rc if rc is not None and rc * 2 <= T else B * T


def cpu_baseline(model_name, P, T, seconds_hint=20.0):
    """The oracle (CPU restatement of the reference path, fp32 eager) timed on this box's host cores on a bounded sample
    of the same workload: same model / sequence shape / MAFED step, batch 4 instead of 32."""
    from oracle import vlpythia_ref as R
    # the GPU box gives one GPU's CPU share (16 threads); os.cpu_count() reports the whole host and oversubscribes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail = cores
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    B = 4
    cfg = R.preset(model_name, num_vision_tokens=P)
    g = torch.Generator().manual_seed(1234)
    sd = {}
    for name, shape in R.param_shapes(cfg):  # HF init distribution; same shapes/names as the GPU run
        if "layernorm" in name or "layer_norm" in name:
            sd[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith("bias"):
            sd[name] = torch.zeros(shape)
        else:
            sd[name] = torch.randn(shape, generator=g) * 0.02
    tsd = {k: v + 1e-3 * torch.randn(v.shape, generator=g) for k, v in sd.items()}
    batch = R.make_batch(cfg, B, T, seed=1235, pad=False)
    tr = R.RefTrainer(cfg, sd, lr=5e-5, accumulate=1, replay_interval=1, warmup_steps=0, total_steps=1000, task_id=1, teacher_sd=tsd,
                      spec=R.DistillSpec(modality="balanced", layer_strategy="discounted", gamma=0.5))
    tw = time.time()
    tr.step(batch, 0, batch)  # warm-up
    print(f"[bench] cpu baseline warm-up step {time.time() - tw:.1f} s on {cores} threads", file=sys.stderr, flush=True)
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < seconds_hint and n < 8):
        tr.step(batch, n + 1, batch)
        n += 1
    dt = time.time() - t0
    out = {"value": round(B * n / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"oracle fp32 CPU, VLPythia-{model_name}+MAFED distill step, batch {B} (of 32), {P} img + {T} txt tokens, {n} timed steps; "
                     f"{cores} threads used of {avail} the box exposes to this process (one GPU's CPU share is 16)"}
    del tr, sd, tsd
    # BASELINE.json configs[0], the reference's own CPU-runnable case, at its full size: VLPythia-160M naive finetune (no distillation),
    # batch 4, 64 image + 16 text tokens
    try:
        cfg0 = R.preset("160m", num_vision_tokens=64)
        sd0 = R.init_weights(cfg0, seed=1234)
        b0 = R.make_batch(cfg0, 4, 16, seed=1235, pad=False)
        tr0 = R.RefTrainer(cfg0, sd0, lr=5e-5, accumulate=1, replay_interval=4, warmup_steps=0, total_steps=1000, task_id=0)
        tr0.step(b0, 0)
        n0, t0 = 0, time.time()
        while n0 < 3 or (time.time() - t0 < 6.0 and n0 < 40):
            tr0.step(b0, n0 + 1)
            n0 += 1
        dt0 = time.time() - t0
        out["configs0"] = {"value": round(4 * n0 / dt0, 3), "unit": "samples/s", "cores": cores, "kind": "port",
                           "sample": f"oracle fp32 CPU, VLPythia-160M naive finetune step (CE only, clip, AdamW), batch 4, 64 img + 16 txt tokens, "
                                     f"{n0} timed steps (BASELINE.json configs[0] at full size)"}
    except Exception as e:
        out["configs0"] = {"value": None, "sample": f"failed: {e}"}
    return out


def _spawn_ranks(n: int) -> int:
    """``--gpus n`` outside a launcher: become the launcher.  Runs before any HIP call in this process (device_count() does
    not initialise the GPU on this image) and hands the child's exit code back."""
    have = torch.cuda.device_count()
    rehearsal = os.environ.get("MAFED_DIST_BACKEND") == "gloo"  # several ranks sharing one GPU over gloo: a dry run of the N > 1 code path
    if have < n and not rehearsal:
        print(f"bench.py: --gpus {n} requested but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def dump_profile(directory, name, kp, rank):
    """The raw records of one KernelProfile (what every `frac` of the line is computed from) -> CSV, so that the numbers can be
    recomputed without re-running: achieved = sum(work) / sum(duration) per kernel tag."""
    if not directory or rank != 0:
        return
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, name), "w") as fp:
        fp.write("# in-library kernel profiler (start/stop events per launch, hipExtLaunchKernelGGL); work = algorithmic flops (gemm / attn tags) or bytes\n")
        fp.write("kernel,work,start_ms,duration_us\n")
        for (tag, work, ms), st in zip(kp.records(), kp.starts_ms):
            fp.write(f"{tag},{work:.6g},{st:.4f},{ms * 1e3:.3f}\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="410m")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img-tokens", type=int, default=256)
    ap.add_argument("--txt-tokens", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true", help="skip the per-kernel roofline steps after the timed region")
    ap.add_argument("--dump-profile", default=None, metavar="DIR", help="write the per-launch records behind `roofline` / `kernels` as CSV (tag, "
                    "algorithmic work, start ms, duration us) for the timed configuration and for the single-stream repeat")
    ap.add_argument("--no-secondary", action="store_true", help="skip the blended-schedule measurement (3 CE + 1 MAFED micro-batches)")
    ap.add_argument("--no-teacher-cache-leg", action="store_true", help="skip the teacher-cache measurement (frozen-teacher states of the whole "
                    "replay memory resident in HBM: 108 GB at 410M / 4000 samples)")
    ap.add_argument("--no-image-leg", action="store_true", help="skip the image-input measurement (CLIP-ViT-L/14 tower in front of the step)")
    ap.add_argument("--gemm-variant", type=int, default=None, help="tuning: mafed_gemm_set_variant value")
    ap.add_argument("--no-pipeline-optimizer", action="store_true", help="AdamW in front of the next forward instead of under it")
    ap.add_argument("--dense-head", action="store_true", help="LM head, CE and the head's gradient GEMMs on all 32 text positions (A/B)")
    ap.add_argument("--no-defer-ln", action="store_true", help="LayerNorm parameter reduction on the dX stream (A/B)")
    ap.add_argument("--no-incremental-norm", action="store_true", help="clip norm in one pass at the start of the optimiser step (A/B)")
    ap.add_argument("--no-overlap", action="store_true", help="single stream (profiling: per-kernel durations without concurrency)")
    ap.add_argument("--no-overlap-teacher", action="store_true", help="teacher forward on the caller's stream (A/B)")
    ap.add_argument("--no-overlap-dw", action="store_true", help="parameter-gradient work on the caller's stream (A/B)")
    ap.add_argument("--contended-backward", action="store_true", help="N = 1 diagnostic: run the backward the way it runs under data parallelism "
                    "(128 x 128 GEMM kernels, weight gradients per product on the side streams) -- what the kernel choice alone costs per GPU")
    ap.add_argument("--dw-group-layers", type=int, default=None, help="layers per grouped weight-gradient launch (0 = one launch per product)")
    ap.add_argument("--reduce-mode", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="N > 1: one all-reduce per gradient bucket, or reduce-scatter + all-gather")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="N > 1: dtype of the gradient buckets on the links")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--hw-queues", type=int, default=None, help="GPU_MAX_HW_QUEUES for this process (default 4; read before the HIP runtime starts)")
    ap.add_argument("--no-dev-kernarg", action="store_true", help="leave HIP_FORCE_DEV_KERNARG at the runtime's default")
    ap.add_argument("--rccl-channels", type=int, default=None, help="N > 1: pin RCCL's channel count (NCCL_MIN/MAX_NCHANNELS)")
    ap.add_argument("--memory-size", type=int, default=4000, help="samples resident in the HBM replay memory (the reference's --memory_size)")
    ap.add_argument("--exact-normaliser", action="store_true", help="N > 1: distillation means over the GLOBAL token counts (one small all-reduce)")
    ap.add_argument("--no-ddp-forecast", action="store_true", help="N = 1: skip the emulated-collectives leg (`ddp_forecast` key)")
    ap.add_argument("--emulate-channels", type=int, default=16, help="N = 1 forecast: workgroups of the stand-in collective kernel (RCCL channels)")
    ap.add_argument("--tile-order", default="auto", choices=["auto", "static", "ticketed"], help="persistent GEMM kernels: auto = per call (static "
                    "on a free chip, ticketed for the backward beside collectives), static / ticketed = everywhere (A/B)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus))

    from mafed_amd import FeatureDistillation, Trainer, VLPythiaConfig, VLPythiaForCausalLM
    from mafed_amd.dist import broadcast_teacher, init_from_env
    from mafed_amd.methods import HBMReplayBuffer
    from mafed_amd.profiler import KernelProfile
    from mafed_amd.runtime_env import runtime_env
    import torch.distributed as dist

    if args.gemm_variant is not None:
        from mafed_amd import _lib
        _lib.load().mafed_gemm_set_variant(args.gemm_variant)
    if args.tile_order != "auto":
        from mafed_amd import _lib
        _lib.load().mafed_gemm_set_variant(720 if args.tile_order == "static" else 721)
    rank, local, world = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (mafed_amd has no CPU path)")
    ranks_joined = dist.get_world_size() if world > 1 else 1
    if ranks_joined != args.gpus:
        raise SystemExit(f"bench.py: {ranks_joined} ranks joined the process group, {args.gpus} requested")
    backend = dist.get_backend() if world > 1 else None
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node exposes {torch.cuda.device_count()}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, P, T = args.batch, args.img_tokens, args.txt_tokens
    cfg = VLPythiaConfig.preset(args.model, num_vision_tokens=P)
    cd = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    gdt = torch.bfloat16 if args.grad_dtype == "bf16" else None

    # synthetic inputs of SURVEY.md section 8d: weights N(0, 0.02) (HF init), teacher = student + N(0, 1e-3)
    student = VLPythiaForCausalLM(cfg, compute_dtype=cd, device=dev, seed=1234)
    opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
    fd._update_model(student)
    g = torch.Generator(device=dev).manual_seed(1237)
    fd.past_model.flat_params.add_(torch.randn(fd.past_model.flat_params.shape, generator=g, device=dev) * 1e-3)
    fd.past_model._shadow_dirty = True
    broadcast_teacher(fd.past_model)
    fd.task_id = 1
    fd.num_vision_tokens = P
    if args.no_defer_ln:
        student.defer_ln_param_reduce = False
    if args.dense_head:
        student.sparse_lm_head = False
    if args.no_overlap:
        student.overlap_param_grads = False
        fd.overlap_teacher = False
    if args.no_overlap_teacher:
        fd.overlap_teacher = False
    if args.no_overlap_dw:
        student.overlap_param_grads = False
    if args.dw_group_layers is not None:
        student.dw_group_layers = args.dw_group_layers
    fd.exact_normaliser = bool(args.exact_normaliser)
    if args.contended_backward:
        student.contended_backward = "128x128"
    # the replay memory holds --memory-size samples (the reference's memory_size = 4000, scripts/run_seed42.sh) resident in HBM: bf16 patch
    # features [n, P, dv] (2.1 GB at 4000 x 256 x 1024) + int64 text tensors; generated on the device in chunks, rank-specific seeds
    n_mem = max(8 * B, args.memory_size)
    gcpu = torch.Generator().manual_seed(1235 + rank)
    ids = torch.randint(1, cfg.vocab_size, (n_mem, T), generator=gcpu)
    labels = torch.full((n_mem, T), -100, dtype=torch.int64)
    labels[:, -4:] = ids[:, -4:]
    gmem = torch.Generator(device=dev).manual_seed(1234 + rank)
    mem = HBMReplayBuffer(B, dev, seed=1236 + rank)
    chunk = 512
    feats = torch.empty(n_mem, P, cfg.vision_hidden_size, dtype=torch.bfloat16, device=dev)
    for lo in range(0, n_mem, chunk):
        hi = min(n_mem, lo + chunk)
        feats[lo:hi] = torch.randn(hi - lo, P, cfg.vision_hidden_size, generator=gmem, device=dev).to(torch.bfloat16)
    mem_samples = {"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels, "patch_embeddings": feats}
    mem.add(mem_samples)
    del feats
    mem_small = {k: v[: 8 * B] for k, v in mem_samples.items()}   # the secondary / image legs keep a small memory of their own
    fd.mem_dataloader = mem
    conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98),
                                 weight_decay=0.01, optim="adamw", warmup_perc=0.1)
    tr = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=1000, ddp=world > 1, pipeline_optimizer=not args.no_pipeline_optimizer,
                 bucket_mb=args.bucket_mb, reduce_mode=args.reduce_mode, grad_dtype=gdt, incremental_norm=not args.no_incremental_norm)
    if args.gemm_variant is not None:
        tr.contention_mode = None   # an explicit kernel choice is not overridden under N > 1
    task_batch = mem.sample()  # dropped by a replay step, as in the reference

    def barrier():
        if world > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"setup done: {args.model} B={B} P={P} T={T} dtype={args.dtype} world={world} backend={backend}")
    def alloc_state():   # (device allocations of torch's caching allocator so far, reserved GB): a step that grows the pool calls hipMalloc,
        ms = torch.cuda.memory_stats(dev)   # which synchronises the device -- the "2x warm-up steps" of the round-3 logs
        return int(ms.get("num_device_alloc", 0)), ms.get("reserved_bytes.all.current", 0) / 1e9

    for i in range(args.warmup):
        a0 = alloc_state()
        tw = time.perf_counter()
        tr.step(task_batch, i)
        torch.cuda.synchronize()
        a1 = alloc_state()
        log(f"warmup step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms (hipMalloc calls {a1[0] - a0[0]}, pool {a0[1]:.2f} -> {a1[1]:.2f} GB)")
    torch.cuda.synchronize()
    if tr.reducer is not None:
        tr.reducer.time_wait = True   # two event records per step on the compute stream (diagnostics of the N > 1 line)
    barrier()
    torch.cuda.synchronize()
    a_timed0 = alloc_state()
    c0 = time.thread_time()
    t0 = time.perf_counter()
    for i in range(args.steps):
        rec = tr.step(task_batch, args.warmup + i)
    t_host = time.perf_counter() - t0  # host enqueue wall time (includes standing behind the full launch queues)
    c_host = time.thread_time() - c0   # CPU time of the enqueueing thread: the host's own work (autograd's backward thread not included)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0  # this rank's own time, before the barrier
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        waits = tr.reducer.exposed_wait_ms() if tr.reducer is not None else []
        tr.reducer.time_wait = False
        mine = torch.tensor([dt_rank / args.steps * 1e3, sum(waits) / max(1, len(waits)), max(waits) if waits else 0.0], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(x[0]), 3) for x in allr], "reducer_wait_ms_mean": [round(float(x[1]), 3) for x in allr],
                    "reducer_wait_ms_max": [round(float(x[2]), 3) for x in allr]}
    dt = float(tmax.item())
    loss = float(rec["loss"])
    a_timed1 = alloc_state()
    log(f"timed region: {args.steps} steps in {dt * 1e3:.1f} ms (host enqueue wall {t_host * 1e3:.1f} ms, CPU {c_host * 1e3:.1f} ms; "
        f"hipMalloc calls inside {a_timed1[0] - a_timed0[0]}), loss {loss:.5f}")
    assert loss == loss, "NaN loss in the timed region"

    # live rooflines: N_PROF more steps of the SAME configuration (same streams / overlap) with every hot kernel launched
    # between a start/stop event pair: per-kernel execution time, as rocprofv3 --kernel-trace would report it.  Not part of
    # `value`.  Every rank takes these steps (they contain the gradient collectives); rank 0 reports.
    N_PROF = 3
    prof_step, prof_alone = None, None
    if not args.no_kernel_profile:
        with KernelProfile() as kp:
            for i in range(N_PROF):
                tr.step(task_batch, args.warmup + args.steps + i)
            torch.cuda.synchronize()
        prof_step = kp.summary()
        dump_profile(args.dump_profile, "kernel_profile_timed.csv", kp, rank)
        # ... and once more with the side streams switched off (dW GEMMs, teacher forward and the optimiser in line on one
        # stream): every kernel then has the chip to itself -- what a per-kernel roofline is read against
        ov = (student.overlap_param_grads, fd.overlap_teacher, tr.pipeline_optimizer)
        tr.join()
        student.overlap_param_grads, fd.overlap_teacher, tr.pipeline_optimizer = False, False, False
        with KernelProfile() as kp1:
            for i in range(N_PROF):
                tr.step(task_batch, args.warmup + args.steps + N_PROF + i)
            torch.cuda.synchronize()
        prof_alone = kp1.summary()
        dump_profile(args.dump_profile, "kernel_profile_no_overlap.csv", kp1, rank)
        student.overlap_param_grads, fd.overlap_teacher, tr.pipeline_optimizer = ov
        barrier()

    # Forecast of the N-GPU step on ONE GPU (no multi-GPU node has ever been available to this build -- this is NOT a scaling measurement):
    # the same step with dist.EmulatedReducer in place of the gradient exchange: per bucket, `--emulate-channels` workgroups stream memory
    # on the reducer's side stream for the time that bucket's all-reduce would take, beside the backward.  Two exchange durations (SURVEY
    # section 5: 2.7 ms with all seven xGMI links busy, 18.6 ms for one ring) x the two ways the backward beside it can run: the persistent
    # kernels in ticketed order, or every GEMM of that backward on the 128 x 128 kernels (MAFED_EPI_NO_PERSISTENT, round 3's choice).
    ddp_forecast = None
    if world == 1 and not args.no_ddp_forecast:
        from mafed_amd.dist import EmulatedReducer
        tr.join()
        rows = []
        n_f = max(5, args.steps // 2)
        for ar_ms in (2.7, 18.6):
            for mode in ("persistent_ticketed", "persistent_static", "128x128"):
                student.zero_grad()
                red = EmulatedReducer(student, ar_ms, channels=args.emulate_channels, bucket_mb=args.bucket_mb)
                trf = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=1000, pipeline_optimizer=not args.no_pipeline_optimizer,
                              incremental_norm=not args.no_incremental_norm, reducer=red)
                trf.contention_mode = {"persistent_ticketed": "ticketed", "persistent_static": None, "128x128": "128x128"}[mode]
                for i in range(3):
                    trf.step(task_batch, 20_000 + i)
                torch.cuda.synchronize()
                red.time_wait = True
                tf0 = time.perf_counter()
                for i in range(n_f):
                    rec_f = trf.step(task_batch, 20_003 + i)
                torch.cuda.synchronize()
                dtf = time.perf_counter() - tf0
                waits = red.exposed_wait_ms()
                trf.join()
                student.grad_ready_hook = None
                student.contended_backward = False
                rows.append({"allreduce_ms": ar_ms, "kernel_mode": mode, "ms_per_step": round(dtf / n_f * 1e3, 3),
                             "reducer_wait_ms": round(sum(waits) / max(1, len(waits)), 3), "final_loss": round(float(rec_f["loss"]), 5)})
                log(f"ddp forecast: all-reduce {ar_ms} ms, {mode}: {dtf / n_f * 1e3:.2f} ms per step, reducer wait {rows[-1]['reducer_wait_ms']:.2f} ms")
        ddp_forecast = {"what": "ONE GPU, emulated gradient exchange (dist.EmulatedReducer: per bucket, `channels` workgroups streaming memory for the "
                                "bucket's share of `allreduce_ms`, on the reducer's stream beside the backward); a forecast of the per-GPU step "
                                "under data parallelism, not a scaling measurement", "channels": args.emulate_channels, "emulated_world": 8,
                        "buckets": len(red.buckets), "MB_per_step": round(red.bytes_per_step / 1e6, 1), "steps": n_f, "rows": rows}
        student.zero_grad()

    # MI355X-only design point, reported BESIDE the headline (never instead of it: the step above runs the teacher forward, as SURVEY 8d
    # defines it): the frozen teacher's distilled hidden states for this rank's whole replay memory stay resident in HBM
    # (FeatureDistillation.build_teacher_cache: fp32 [layers, samples, S, h], filled once per task by the same kernels), a replay step
    # gathers its 32 samples' rows instead of running the teacher forward
    teacher_cache = None
    if not args.no_teacher_cache_leg:
        tr.join()
        need = (cfg.num_hidden_layers - 1) * len(mem) * (P + T) * cfg.hidden_size * 4
        free, _tot = torch.cuda.mem_get_info(dev)
        if need + (24 << 30) <= free:
            info = fd.build_teacher_cache(mem)
            ntc = max(5, args.steps // 2)
            for i in range(3):
                tr.step(task_batch, 10_000 + i)
            torch.cuda.synchronize()
            barrier()
            ttc = time.perf_counter()
            for i in range(ntc):
                rec_tc = tr.step(task_batch, 10_003 + i)
            torch.cuda.synchronize()
            barrier()
            dtc = time.perf_counter() - ttc
            t4 = torch.tensor([dtc], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t4, op=dist.ReduceOp.MAX)
            dtc = float(t4.item())
            tr.join()
            teacher_cache = {"metric": "train samples/s, the headline step with the frozen teacher's hidden states of the whole replay memory "
                                       "cached in HBM (a gather instead of the teacher forward; bit-identical teacher states)",
                             "value": round(ntc * B * world / dtc, 3), "unit": "samples/s", "steps": ntc, "ms_per_step": round(dtc / ntc * 1e3, 3),
                             "cache_GB_per_gpu": round(info["GB"], 1), "cached_samples_per_gpu": info["samples"], "fill_seconds": round(info["seconds"], 2),
                             "final_loss": round(float(rec_tc["loss"]), 5)}
            assert teacher_cache["final_loss"] == teacher_cache["final_loss"], "NaN loss in the teacher-cache leg"
            log(f"teacher-cache leg: {ntc} steps in {dtc * 1e3:.1f} ms ({info['GB']:.1f} GB cached in {info['seconds']:.1f} s)")
            fd.drop_teacher_cache()
            torch.cuda.empty_cache()
        else:
            teacher_cache = {"value": None, "note": f"skipped: {need / 1e9:.0f} GB of teacher states do not fit beside {(_tot - free) / 1e9:.0f} GB in use"}

    # secondary metric: the reference's blended schedule -- bs 16 x accumulate 4, replay_interval 4: per optimiser
    # step three plain-CE micro-batches on task data and one MAFED micro-batch on memory data (scripts/run_seed42.sh:51-70)
    secondary = None
    if not args.no_secondary:
        tr.join()
        B2 = 16
        mem2 = HBMReplayBuffer(B2, dev, seed=2236 + rank)
        mem2.add(mem_small)
        fd.mem_dataloader = mem2
        fd.batch_size = B2
        conf2 = types.SimpleNamespace(accumulate_grad_batches=4, replay_interval=4, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98),
                                      weight_decay=0.01, optim="adamw", warmup_perc=0.1)
        student.zero_grad()
        tr2 = Trainer(student, fd, conf2, task_id=1, n_batches_per_epoch=1000, ddp=world > 1, pipeline_optimizer=not args.no_pipeline_optimizer,
                      bucket_mb=args.bucket_mb, reduce_mode=args.reduce_mode, grad_dtype=gdt, incremental_norm=not args.no_incremental_norm)
        # current-task micro-batches (synthetic, resident in HBM) WITHOUT the loader's label-row hint: a task DataLoader's collate does not
        # attach it, so the three plain-CE micro-batches run the dense LM head; the memory micro-batch keeps the hint (row-sparse head)
        task2 = [{k: v for k, v in mem2._draw().items() if k != "max_label_rows"} for _ in range(4)]
        n_opt = max(3, args.steps // 4)
        for i in range(8):                          # two untimed optimiser steps
            tr2.step(task2[i % 4], i)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for i in range(4 * n_opt):
            rec2 = tr2.step(task2[i % 4], 8 + i)
        torch.cuda.synchronize()
        barrier()
        dt2 = time.perf_counter() - t1
        t2 = torch.tensor([dt2], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        dt2 = float(t2.item())
        tr2.join()
        fl2 = B2 * (3 * algorithmic_flops_per_sample(cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size, P, T, cfg.vision_hidden_size, distill=False)
                    + algorithmic_flops_per_sample(cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size, P, T, cfg.vision_hidden_size, distill=True))
        # executed flops, as in the headline: the MAFED micro-batch carries the hint and runs the row-sparse head, the three task micro-batches do not
        hr2 = lm_head_rows(B2, T, mem2.max_label_rows if mem2.attach_label_hint else None, cd == torch.bfloat16, student.sparse_lm_head)
        fl2 -= 3.0 * 2.0 * cfg.hidden_size * cfg.vocab_size * (B2 * T - hr2)
        secondary = {"metric": "train samples/s, reference schedule: bs16 x accum4, 3 plain-CE + 1 MAFED micro-batch per optimiser step",
                     "value": round(4 * B2 * n_opt * world / dt2, 3), "unit": "samples/s", "optimizer_steps": n_opt,
                     "ms_per_optimizer_step": round(dt2 / n_opt * 1e3, 3), "samples_per_optimizer_step": 4 * B2 * world,
                     "mfma_frac_whole_step": round(fl2 * n_opt / dt2 / 1e12 / PEAK_BF16_TFLOPS, 4), "final_loss": round(float(rec2["loss"]), 5)}
        log(f"secondary: {n_opt} optimiser steps in {dt2 * 1e3:.1f} ms")

    # third leg: the same all-distill step fed with IMAGES (pixel_values [B,3,224,224] -> native frozen CLIP-ViT-L/14 tower -> step), the
    # data path of upstream's training loop; the tower runs once per step (upstream: twice, distillation.py:91,222).  Random tower weights.
    image_leg = None
    if not args.no_image_leg and cfg.vision_hidden_size == 1024 and P == 256:
        from mafed_amd.vision import ClipVisionConfig, ClipVisionTower
        tr.join()
        student.zero_grad()
        tower = ClipVisionTower(ClipVisionConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                                                 image_size=224, patch_size=14), compute_dtype=cd, device=dev)
        gt = torch.Generator(device=dev).manual_seed(4321)
        with torch.no_grad():
            for prm in tower.parameters():
                prm.copy_(torch.randn(prm.shape, generator=gt, device=dev) * 0.02)
        tower._derived = None
        student.vision_encoder = tower
        fd.past_model.vision_encoder = tower
        img_batches = []
        for j in range(2):
            sl = slice(j * B, (j + 1) * B)
            img_batches.append({"input_ids": ids[sl].to(dev), "attention_mask": torch.ones(B, T, dtype=torch.int64, device=dev), "labels": labels[sl].to(dev),
                                "pixel_values": torch.randn(B, 3, 224, 224, generator=gt, device=dev).to(cd)})

        class _Cycle:   # memory loader handing out image batches (fresh dict per draw: replay() adds keys to it)
            def __init__(self):
                self.i = 0
            def __iter__(self):
                return self
            def __next__(self):
                self.i += 1
                return dict(img_batches[self.i % 2])
        fd.mem_dataloader = _Cycle()
        fd.batch_size = B
        tr3 = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=1000, ddp=world > 1, pipeline_optimizer=not args.no_pipeline_optimizer,
                      bucket_mb=args.bucket_mb, reduce_mode=args.reduce_mode, grad_dtype=gdt, incremental_norm=not args.no_incremental_norm)
        n3 = max(5, args.steps // 2)
        for i in range(3):
            tr3.step(task_batch, i)
        torch.cuda.synchronize()
        barrier()
        t3 = time.perf_counter()
        for i in range(n3):
            rec3 = tr3.step(task_batch, 3 + i)
        torch.cuda.synchronize()
        barrier()
        dt3 = time.perf_counter() - t3
        t3t = torch.tensor([dt3], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t3t, op=dist.ReduceOp.MAX)
        dt3 = float(t3t.item())
        tr3.join()
        image_leg = {"metric": "train samples/s, the headline step fed with images: pixel_values [B,3,224,224] bf16 -> frozen CLIP-ViT-L/14 tower "
                               "(23 of 24 layers, 257 tokens, once per step for student + teacher) -> MAFED step",
                     "value": round(n3 * B * world / dt3, 3), "unit": "samples/s", "steps": n3, "ms_per_step": round(dt3 / n3 * 1e3, 3),
                     "final_loss": round(float(rec3["loss"]), 5)}
        assert image_leg["final_loss"] == image_leg["final_loss"], "NaN loss in the image-input leg"
        log(f"image-input leg: {n3} steps in {dt3 * 1e3:.1f} ms")

    if rank == 0:
        samples = args.steps * B * world
        value = samples / dt
        flops_step = algorithmic_flops_per_sample(cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size, P, T, cfg.vision_hidden_size) * B
        hint = getattr(mem, "max_label_rows", None) if getattr(mem, "attach_label_hint", False) else None
        head_rows = lm_head_rows(B, T, hint, cd == torch.bfloat16, student.sparse_lm_head)
        flops_exec = flops_step - 3.0 * 2.0 * cfg.hidden_size * cfg.vocab_size * (B * T - head_rows)
        roof, kernels = None, None

        def merged(prof, tags):   # one figure over several profiler tags: sum(work) / sum(duration)
            rows = [prof[t] for t in tags if prof and t in prof]
            if not rows:
                return None
            m = {"launches": sum(r["launches"] for r in rows), "total_ms": sum(r["total_ms"] for r in rows), "work": sum(r["work"] for r in rows)}
            m["avg_us"] = m["total_ms"] * 1e3 / m["launches"]
            m["achieved"] = m["work"] / (m["total_ms"] * 1e-3) / 1e12
            m["frac"] = m["achieved"] / PEAK_BF16_TFLOPS
            return m

        # dominant kernel = the bf16 MFMA GEMM: the persistent ping-pong kernels (tag gemm_pp: gemm_pp_kernel / gemm_z_kernel) plus
        # the shapes they do not tile, which stay on gemm_bf16_glds_kernel (tag gemm_bf16)
        GEMM_TAGS = ("gemm_pp", "gemm_bf16")
        g_alone, g_step = merged(prof_alone, GEMM_TAGS), merged(prof_step, GEMM_TAGS)
        if g_alone or g_step:
            traffic, tsrc = None, None
            for cand in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                try:  # HBM-side bytes per launch of this kernel from the committed PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE)
                    with open(os.path.join(ROOT, "profiles", cand)) as fp:
                        traffic = json.load(fp)["hbm_MB_per_launch"] * 1e6
                    tsrc = f"profiles/{cand} (separate rocprofv3 --pmc passes of this command; not measured in this run)"
                    break
                except Exception:
                    pass
            # `frac` is the kernels' own figure: every launch with the chip to itself (side streams off, same step, same shapes).  Inside
            # the timed configuration several kernels share the CUs, which stretches each launch: that figure is `frac_in_step`.
            gm = g_alone or g_step
            roof = {"bound": "mfma", "achieved": round(gm["achieved"], 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(gm["frac"], 4), "traffic": traffic, "traffic_source": tsrc,
                    "kernel": "bf16 MFMA GEMM: gemm_pp_kernel + gemm_z_kernel (persistent, tag gemm_pp) and gemm_bf16_glds_kernel (tag gemm_bf16)",
                    "mode": f"sum(2MNK) / sum(kernel execution time) over {N_PROF} steps with the side streams off (each kernel has the chip to "
                            "itself), start/stop events on each launch = rocprofv3 --kernel-trace durations; `frac_in_step` = the same over "
                            f"{N_PROF} steps of the timed configuration (kernels of several streams share the chip)",
                    "records": "profiles/r04_kernel_profile_no_overlap.csv / r04_kernel_profile_timed.csv (this command with --dump-profile); "
                               "rocprofv3 --kernel-trace --stats of the same command: profiles/r04_bench_kernel_stats.csv",
                    "launches_per_step": round(gm["launches"] / N_PROF, 1), "avg_launch_us": round(gm["avg_us"], 2),
                    "avg_gflop_per_launch": round(gm["work"] / gm["launches"] / 1e9, 3)}
            if g_step and g_alone:
                roof["achieved_in_step"] = round(g_step["achieved"], 2)
                roof["frac_in_step"] = round(g_step["frac"], 4)
                roof["avg_launch_us_in_step"] = round(g_step["avg_us"], 2)
            for t in GEMM_TAGS:
                r1 = prof_alone.get(t) if prof_alone else None
                if r1:
                    roof[f"{t}_share"] = {"launches_per_step": round(r1["launches"] / N_PROF, 1), "tflops": round(r1["achieved"], 1),
                                          "flop_share": round(r1["work"] / gm["work"], 4)} if g_alone else None
            kernels = []
            for tag in ("gemm_pp", "gemm_bf16", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "layernorm_fwd", "layernorm_bwd", "ce_fwd", "ce_bwd", "distill_fwd",
                        "adamw", "gradnorm", "embed_concat_fwd", "embed_concat_bwd", "cast", "colsum", "layernorm_bwd_reduce"):
                a = prof_step.get(tag)
                if not a:
                    continue
                e = {"kernel": tag, "bound": a["bound"], "launches_per_step": round(a["launches"] / N_PROF, 1), "avg_us": round(a["avg_us"], 2),
                     "ms_per_step": round(a["total_ms"] / N_PROF, 3),
                     ("gflop_per_launch" if a["bound"] == "mfma" else "MB_per_launch"): round(a["work"] / a["launches"] / (1e9 if a["bound"] == "mfma" else 1e6), 3),
                     "achieved": round(a["achieved"], 1), "unit": a["unit"], "peak": a["peak"], "frac": round(a["frac"], 4)}
                a1 = prof_alone.get(tag) if prof_alone else None
                if a1:
                    e["avg_us_no_overlap"] = round(a1["avg_us"], 2)
                    e["frac_no_overlap"] = round(a1["frac"], 4)
                kernels.append(e)
        out = {"metric": "train samples/sec VLPythia-410M+MAFED, 256img+32txt tok, bs=32, 1/2/4/8 GPU" if args.model == "410m" else
               f"train samples/sec VLPythia-{args.model}+MAFED", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"VLPythia-{args.model} + MAFED distill step (student fwd+bwd, frozen-teacher fwd, replay CE, "
                          f"{cfg.num_hidden_layers - 1}-layer per-modality MSE, clip 2.0, AdamW), {P} img + {T} txt tokens",
                          "global_batch": B * world, "per_gpu_batch": B, "seq_len": P + T, "parallelism": f"dp{world}",
                          "random_init_weights": True},
               "ranks_joined": ranks_joined, "dist_backend": backend, "runtime_env": runtime_env(),
               "gemm_tile_order": {"auto": "per call: static on a free chip, ticketed beside collectives"}.get(args.tile_order, args.tile_order),
               "replay_memory": {"samples": len(mem), "HBM_MB": round(sum(v.numel() * v.element_size() for v in mem.data.values() if v is not None) / 1e6, 1)},
               "step_tflops_algorithmic": round(flops_step / 1e12, 3),
               "mfma_frac_whole_step": round(flops_exec * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "final_loss": round(loss, 5),
               # the enqueueing thread's CPU time per step (its own work) beside the wall time it spent in the loop (which includes standing
               # behind full launch queues): host-bound would be host_enqueue_cpu_ms_per_step ~ ms_per_step
               "host_enqueue_cpu_ms_per_step": round(c_host / args.steps * 1e3, 3), "host_enqueue_wall_ms_per_step": round(t_host / args.steps * 1e3, 3),
               "device_allocs_in_timed_region": a_timed1[0] - a_timed0[0]}
        if head_rows != B * T:
            # SURVEY 8d counts the LM head on all T text positions; the row-sparse head runs it (forward + both gradient GEMMs) on the
            # rows that carry a label -- `mfma_frac_whole_step` is priced on the flops actually executed
            out["lm_head"] = {"rows_per_step": head_rows, "of": B * T, "step_tflops_executed": round(flops_exec / 1e12, 3)}
        if world > 1:
            out["grad_exchange"] = {"mode": args.reduce_mode, "dtype": args.grad_dtype, "bucket_mb": args.bucket_mb,
                                    "MB_per_step": round(tr.reducer.bytes_per_step / 1e6, 1), "buckets": len(tr.reducer.buckets),
                                    "exact_normaliser": bool(args.exact_normaliser),
                                    # the collectives' CU budget: RCCL runs one workgroup per channel; unset = RCCL's own choice for the topology
                                    "rccl_channels": {"NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                                                      "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS")},
                                    # reducer_wait = time the compute stream stood still in GradReducer.wait() (0 = exchange hidden under backward)
                                    "per_rank": per_rank}
        if roof:
            out["roofline"] = roof
        if kernels:
            out["kernels"] = kernels
        if secondary:
            out["secondary"] = secondary
        if image_leg:
            out["image_input"] = image_leg
        if teacher_cache:
            out["teacher_cache"] = teacher_cache
        if ddp_forecast:
            out["ddp_forecast"] = ddp_forecast
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.model, P, T)
            except Exception as e:  # the CPU leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
