"""Headline benchmark: train samples/s of the MAFED distill step, VLPythia-410M, 256 image + 32 text tokens,
batch 32 per GPU (BASELINE.json metric / configs[2]); N = 1, 2, 4, 8 GPUs of one node, one process per GPU over RCCL.

A "step" = one full MAFED optimiser step on one synthetic memory batch resident in HBM: student forward with hidden
states + frozen-teacher forward + replay cross-entropy + (L-1)-layer per-modality distillation MSE + backward +
global-norm clip + AdamW (accumulate 1, replay on every step: SURVEY.md section 8d "primary").  Nothing is skipped or
cached inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 410m] [--batch 32] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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


def cpu_baseline(model_name, P, T, seconds_hint=20.0):
    """The oracle (CPU restatement of the reference path, fp32 eager) timed on this box's host cores on a bounded sample
    of the same workload: same model / sequence shape / MAFED step, batch 4 instead of 32."""
    from oracle import vlpythia_ref as R
    # the GPU box gives one GPU's CPU share (16 threads); os.cpu_count() reports the whole host and oversubscribes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
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
    return {"value": round(B * n / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 CPU, VLPythia-{model_name}+MAFED distill step, batch {B} (of 32), {P} img + {T} txt tokens, {n} timed steps"}


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
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--graphs", action="store_true", help="replay the step from a hipGraph (measured slower than eager launches on ROCm 7: 42.9 vs 39.7 ms)")
    ap.add_argument("--no-graphs", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gemm-variant", type=int, default=None, help="tuning: mafed_gemm_set_variant value")
    ap.add_argument("--no-pipeline-optimizer", action="store_true", help="AdamW in front of the next forward instead of under it")
    ap.add_argument("--no-overlap", action="store_true", help="single stream (profiling: per-kernel durations without concurrency)")
    args = ap.parse_args()

    from mafed_amd import FeatureDistillation, Trainer, VLPythiaConfig, VLPythiaForCausalLM, ops
    from mafed_amd.dist import broadcast_teacher, init_from_env
    from mafed_amd.methods import HBMReplayBuffer
    import torch.distributed as dist

    if args.gemm_variant is not None:
        from mafed_amd import _lib
        _lib.load().mafed_gemm_set_variant(args.gemm_variant)
    rank, local, world = init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (mafed_amd has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, P, T = args.batch, args.img_tokens, args.txt_tokens
    cfg = VLPythiaConfig.preset(args.model, num_vision_tokens=P)
    cd = torch.bfloat16 if args.dtype == "bf16" else torch.float32

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
    if args.no_overlap:
        student.overlap_param_grads = False
        fd.overlap_teacher = False
    n_mem = 8 * B
    gcpu = torch.Generator().manual_seed(1235 + rank)
    ids = torch.randint(1, cfg.vocab_size, (n_mem, T), generator=gcpu)
    labels = torch.full((n_mem, T), -100, dtype=torch.int64)
    labels[:, -4:] = ids[:, -4:]
    mem = HBMReplayBuffer(B, dev, seed=1236 + rank)
    mem.add({"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels,
             "patch_embeddings": torch.randn(n_mem, P, cfg.vision_hidden_size, generator=torch.Generator().manual_seed(1234 + rank))})
    fd.mem_dataloader = mem
    conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98),
                                 weight_decay=0.01, optim="adamw", warmup_perc=0.1)
    tr = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=1000, ddp=world > 1, use_graphs=args.graphs and not args.no_graphs,
                 pipeline_optimizer=not args.no_pipeline_optimizer)
    task_batch = mem.sample()  # dropped by a replay step, as in the reference

    def barrier():
        if world > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"setup done: {args.model} B={B} P={P} T={T} dtype={args.dtype} world={world}")
    for i in range(args.warmup):
        tw = time.perf_counter()
        tr.step(task_batch, i)
        torch.cuda.synchronize()
        log(f"warmup step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms")
    torch.cuda.synchronize()
    barrier()
    log(f"graph replay: {bool(tr._graphs)}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        rec = tr.step(task_batch, args.warmup + i)
    t_host = time.perf_counter() - t0  # host enqueue time (the GPU runs behind it)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss = float(rec["loss"])
    log(f"timed region: {args.steps} steps in {dt * 1e3:.1f} ms (host enqueue {t_host * 1e3:.1f} ms), loss {loss:.5f}")
    assert loss == loss, "NaN loss in the timed region"

    # live roofline of the dominant kernel: two more steps, launched eagerly so that every bf16 MFMA GEMM launch can be
    # bracketed by HIP events on its own stream (same kernels, same streams/overlap as the replayed graph; not part of `value`)
    # (every rank takes these steps -- they contain the gradient all-reduce -- but only rank 0 records events)
    ev_alone = None
    if not args.no_gemm_events:
        graphs_on = tr.use_graphs
        tr.use_graphs = False
        ops.GEMM_EVENTS = [] if rank == 0 else None
        for i in range(2):
            tr.step(task_batch, args.warmup + args.steps + i)
        torch.cuda.synchronize()
        ev_overlapped, ops.GEMM_EVENTS = ops.GEMM_EVENTS, ([] if rank == 0 else None)
        # ... and two steps with the side streams switched off (dW GEMMs and the teacher forward in line on the main
        # stream): every GEMM then has the chip to itself, which is the number a per-kernel roofline should be read against
        ov = (student.overlap_param_grads, fd.overlap_teacher)
        student.overlap_param_grads, fd.overlap_teacher = False, False
        for i in range(2):
            tr.step(task_batch, args.warmup + args.steps + 2 + i)
        torch.cuda.synchronize()
        student.overlap_param_grads, fd.overlap_teacher = ov
        ev_alone, ops.GEMM_EVENTS = ops.GEMM_EVENTS, ev_overlapped
        tr.use_graphs = graphs_on
        barrier()
    if rank == 0:
        samples = args.steps * B * world
        value = samples / dt
        flops_step = algorithmic_flops_per_sample(cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size, P, T, cfg.vision_hidden_size) * B
        roof = None
        if ops.GEMM_EVENTS:
            ev = ops.GEMM_EVENTS
            ops.GEMM_EVENTS = None
            ms = sum(a.elapsed_time(b) for a, b, _ in ev)
            fl = sum(f for _, _, f in ev)
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            try:  # HBM-side bytes per launch of this kernel from the committed PMC run (FETCH_SIZE x2-corrected + WRITE_SIZE)
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fp:
                    traffic = json.load(fp)["hbm_MB_per_launch"] * 1e6
            except Exception:
                pass
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "kernel": "gemm_bf16_glds_kernel (every bf16 MFMA GEMM launch of 2 more steps run right after the timed region)",
                    "launches": len(ev), "avg_launch_us": round(ms * 1e3 / len(ev), 2), "avg_gflop_per_launch": round(fl / len(ev) / 1e9, 3),
                    "note": "launch durations overlap: dW GEMMs and the teacher forward run on side streams"}
            if ev_alone:
                ms1 = sum(a.elapsed_time(b) for a, b, _ in ev_alone)
                ach1 = sum(f for _, _, f in ev_alone) / (ms1 * 1e-3) / 1e12
                roof["achieved_no_overlap"] = round(ach1, 2)
                roof["frac_no_overlap"] = round(ach1 / PEAK_BF16_TFLOPS, 4)
                roof["avg_launch_us_no_overlap"] = round(ms1 * 1e3 / len(ev_alone), 2)
        out = {"metric": "train samples/sec VLPythia-410M+MAFED, 256img+32txt tok, bs=32, 1/2/4/8 GPU" if args.model == "410m" else
               f"train samples/sec VLPythia-{args.model}+MAFED", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"VLPythia-{args.model} + MAFED distill step (student fwd+bwd, frozen-teacher fwd, replay CE, "
                          f"{cfg.num_hidden_layers - 1}-layer per-modality MSE, clip 2.0, AdamW), {P} img + {T} txt tokens",
                          "global_batch": B * world, "per_gpu_batch": B, "seq_len": P + T, "parallelism": f"dp{world}",
                          "random_init_weights": True},
               "step_tflops_algorithmic": round(flops_step / 1e12, 3),
               "mfma_frac_whole_step": round(flops_step * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "final_loss": round(loss, 5)}
        if roof:
            out["roofline"] = roof
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
