"""The in-library kernel profiler (mafed_prof_*): tags, algorithmic work and plausible execution times of the launches made
while a profile is open; launches outside a profile are untouched."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_profile_records_kernel_execution_times():
    from mafed_amd import ops
    from mafed_amd.profiler import KernelProfile
    g = torch.Generator(device=DEV).manual_seed(0)
    M, N, K = 2304, 1024, 1024
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = torch.randn(N, K, device=DEV, generator=g).to(torch.bfloat16)
    x = torch.randn(M, N, device=DEV, generator=g)
    w1, b1 = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    ref = ops.gemm(A, W, False, True)                      # outside a profile: plain launch
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with KernelProfile(max_records=64) as kp:
        e0.record()
        for _ in range(5):
            out = ops.gemm(A, W, False, True)
            y, _, _, _ = ops.layernorm_fwd(x, w1, b1, None, None, 1e-5, torch.bfloat16)
        e1.record()
    recs = kp.records()
    assert [r[0] for r in recs] == ["gemm_bf16", "layernorm_fwd"] * 5
    assert torch.equal(out, ref), "a profiled launch runs the same kernel on the same arguments"
    summ = kp.summary()
    gm, ln = summ["gemm_bf16"], summ["layernorm_fwd"]
    assert gm["launches"] == 5 and ln["launches"] == 5
    assert gm["work"] == 5 * 2.0 * M * N * K
    assert ln["work"] == 5 * M * N * (4.0 + 2.0)
    bracket_ms = e0.elapsed_time(e1)
    assert all(ms > 0 for _, _, ms in recs)
    # execution times cannot exceed the stream bracket around them, and they are kernel-sized (us, not ms)
    assert gm["total_ms"] + ln["total_ms"] <= bracket_ms * 1.05 + 0.05, (gm["total_ms"], ln["total_ms"], bracket_ms)
    assert 2.0 < gm["avg_us"] < 500.0 and 0.5 < ln["avg_us"] < 200.0, (gm["avg_us"], ln["avg_us"])
    assert 0.0 < gm["frac"] < 1.0 and gm["bound"] == "mfma" and ln["bound"] == "hbm" and 0.0 < ln["frac"] < 1.0


def test_profile_capacity_degrades_to_plain_launches():
    from mafed_amd import ops
    from mafed_amd.profiler import KernelProfile
    x = torch.randn(64, 256, device=DEV)
    with KernelProfile(max_records=2) as kp:
        for _ in range(4):
            y = ops.cast(x, torch.bfloat16)
    assert len(kp.records()) == 2
    assert torch.equal(y, x.to(torch.bfloat16))


def test_overlapping_profiles_are_rejected():
    """The event pool and record list are process-wide: a second profile opened while one is running would hand out the
    first one's events, so mafed_prof_begin refuses it (and the open profile keeps working)."""
    from mafed_amd import ops
    from mafed_amd.profiler import KernelProfile
    x = torch.randn(64, 256, device=DEV)
    with KernelProfile(max_records=8) as outer:
        ops.cast(x, torch.bfloat16)
        with pytest.raises(RuntimeError, match="already open"):
            with KernelProfile(max_records=8):
                pass
        ops.cast(x, torch.bfloat16)
    assert [r[0] for r in outer.records()] == ["cast", "cast"]
    with KernelProfile(max_records=8) as again:      # closed profiles can be followed by new ones
        ops.cast(x, torch.bfloat16)
    assert len(again.records()) == 1


def test_persistent_gemm_launches_carry_their_own_tag():
    from mafed_amd import ops
    from mafed_amd.profiler import KernelProfile
    g = torch.Generator(device=DEV).manual_seed(0)
    A = torch.randn(9216, 1024, device=DEV, generator=g).to(torch.bfloat16)
    W = torch.randn(4096, 1024, device=DEV, generator=g).to(torch.bfloat16)
    with KernelProfile(max_records=8) as kp:
        ops.gemm(A, W, False, True)
    (tag, work, ms), = kp.records()
    assert tag == "gemm_pp" and work == 2.0 * 9216 * 4096 * 1024 and ms > 0
