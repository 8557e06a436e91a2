"""CPU: the hand-derived tables of the persistent ping-pong GEMM kernel (mafed_amd/csrc/gemm_pp.hip) -- fragment index maps,
LDS bank conflicts and the counted-vmcnt DMA schedule -- checked by the model in tools/pp_schedule_check.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import pp_schedule_check as P  # noqa: E402

CFGS = P.all_cfgs()
IDS = [f"{c.name}-A{'ks' if c.a_ks else 'kc'}-B{'ks' if c.b_ks else 'kc'}-{'bf16' if c.pair else 'f32'}" for c in CFGS]


@pytest.mark.parametrize("c", CFGS, ids=IDS)
def test_fragment_index_maps(c):
    P.check_index_maps(c)


@pytest.mark.parametrize("c", CFGS, ids=IDS)
def test_bank_conflicts(c):
    worst = P.check_bank_conflicts(c)
    for key, ways in worst.items():
        # the one accepted conflict: transposing reads of a [k][n] B image whose columns are pair-mapped for 16-byte bf16 stores
        allowed = 2 if (key == "B:tr" and c.pair) else 1
        assert ways <= allowed, (key, ways)


@pytest.mark.parametrize("c", CFGS, ids=IDS)
@pytest.mark.parametrize("nst,extra", [(9, 0), (9, 9), (16, 20), (18, 18), (18, 40)])
def test_dma_schedule_has_no_raw_or_war_hazard(c, nst, extra):
    assert P.check_schedule(c, nst=nst, extra_epilogue_ops=extra) == []


@pytest.mark.parametrize("c", CFGS, ids=IDS)
@pytest.mark.parametrize("nst,extra", [(9, 0), (18, 0), (9, 9), (16, 20)])
def test_dma_schedule_with_the_ticket_atomic(c, nst, extra):
    """Ticketed order: one more operation (the returning atomic) in the ticket wave's epilogue, one more tolerated by its first wait
    behind the epilogue -- also when the epilogue issues exactly NST stores and nothing else."""
    assert P.check_schedule(c, nst=nst, extra_epilogue_ops=extra, ticket=True) == []


def test_checker_detects_the_relaxed_wait_without_its_atomic(monkeypatch):
    """The +1 is only sound because the atomic is really in the queue: relax the wait in a launch whose epilogue does not carry it."""
    c = P.Cfg("144x256", False, False, True)
    real = P.waits_of
    # (wave 2: its youngest piece in front of the epilogue feeds phase 0 of the next K-tile -- wave 7's feeds phase 1, which is why the
    #  kernel's ticket wave has a whole interval of slack even without the atomic)
    monkeypatch.setattr(P, "TICKET_WAVE", 2)
    monkeypatch.setattr(P, "waits_of", lambda c, wave, post, nst: real(c, wave, post, nst, ticket=True))
    assert P.check_schedule(c, nst=9, extra_epilogue_ops=0, ticket=False) != []


def test_checker_detects_a_loosened_wait(monkeypatch):
    c = P.Cfg("144x256", False, False, True)
    monkeypatch.setattr(P, "waits_of", lambda c, wave, post, nst: {1: 6, 2: 5})
    assert P.check_schedule(c) != []


ZCFGS = P.z_all()


@pytest.mark.parametrize("c", ZCFGS, ids=[f"256x256-A{'ks' if c.a_ks else 'kc'}-B{'ks' if c.b_ks else 'kc'}-{'bf16' if c.pair else 'f32'}" for c in ZCFGS])
def test_z_kernel_index_maps_and_bank_conflicts(c):
    worst = P.z_check(c)
    for key, ways in worst.items():
        allowed = 2 if (key == "B:tr" and c.pair) else 1
        assert ways <= allowed, (key, ways)


@pytest.mark.timeout(600)
def test_ticket_register_is_left_alone_in_the_built_isa():
    """The ticketed tile order of gemm_pp.hip lands a returning atomic in v255 up to a microsecond after the asm statement: the ISA of
    every instantiation must not touch v252 - v255 anywhere else (tools/pp_ticket_audit.py), the synchronous form of the weight-gradient
    kernel must carry its own wait, and no kernel may have scratch."""
    import shutil
    import pp_ticket_audit as T
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    rows = T.audit(T.compile_isa())
    assert len(rows) >= 4
    for r in rows:
        assert not r["problems"], (r["kernel"], r["problems"])
        assert r["atomics"] >= 2 and (r["sync"] or r["parks"] >= 2)
