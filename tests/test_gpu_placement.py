"""-m gpu: field placement (tse_init times a streaming write into up to TSE_PLACEMENT field-sized chunks and gives the five fastest
the roles T, Qdp(1), Qdp(2), B, C).  Pure placement: the state after the same steps is the same bits with and
without it, the report names distinct chunks, and T is the fastest-writing one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(monkeypatch, placement):
    import torch
    from transport_se_amd.driver import PrimRun
    if placement is None:
        monkeypatch.delenv("TSE_PLACEMENT", raising=False)
    else:
        monkeypatch.setenv("TSE_PLACEMENT", placement)
    run = PrimRun(30, 35, test_case=1, device=0, torch_mod=torch)   # scratch fields of 1.7 GB: above the 1 GiB threshold
    run.run(3)                                                      # one rsplit cycle: three tracer steps and a remap
    run.hip.synchronize()
    rep = run.hip.placement()
    q = run.hip.fetch("qdp", (2, run.nelem, 35, 72, 4, 4)).copy()
    run.close()
    return rep, q


def test_placement_moves_no_bit_and_reports_its_choice(monkeypatch):
    off, q0 = _run(monkeypatch, "0")
    assert off["candidates"] == 0
    on, q1 = _run(monkeypatch, None)
    assert on["candidates"] == 7 and len(set(on["chosen"])) == 5 and all(0 <= i < 7 for i in on["chosen"])
    bw = on["write_GBs"]
    assert all(1000 < b < 9000 for b in bw), bw
    ranked = [bw[i] for i in on["chosen"]]
    assert ranked == sorted(ranked, reverse=True) and ranked[0] == max(bw) and ranked[-1] >= sorted(bw)[2]
    assert np.array_equal(q0.view(np.uint64), q1.view(np.uint64))
    six, q2 = _run(monkeypatch, "6")
    assert six["candidates"] == 6
    assert np.array_equal(q0.view(np.uint64), q2.view(np.uint64))
