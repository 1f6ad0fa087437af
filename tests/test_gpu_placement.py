"""-m gpu: field placement (tse_init tries field-sized chunks until three of those it holds take a streaming write at
TSE_PLACEMENT_GOOD, and gives the five fastest the roles T, Qdp(1), Qdp(2), B, C).  Pure placement: the state after
the same steps is the same bits with and without it, and the report is consistent with the rule."""
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
    assert off["tried"] == 0
    on, q1 = _run(monkeypatch, None)
    bw, sel = on["write_GBs"], on["chosen"]
    assert 5 <= on["tried"] <= 20 and len(bw) == on["tried"] and len(set(sel)) == 5 and all(1000 < b < 9000 for b in bw), on
    ranked = [bw[i] for i in sel]
    assert ranked == sorted(ranked, reverse=True), on                  # T gets the fastest, then Qdp(1), Qdp(2), B, C
    assert on["tried"] == 20 or ranked[2] >= 6000, on                  # it stopped early only with three good chunks in hand
    assert np.array_equal(q0.view(np.uint64), q1.view(np.uint64))
    monkeypatch.setenv("TSE_PLACEMENT_GOOD", "1e9")    # never good enough: the whole budget is used
    nine, q2 = _run(monkeypatch, "9")
    assert nine["tried"] == 9
    assert np.array_equal(q0.view(np.uint64), q2.view(np.uint64))


def test_placement_under_memory_pressure_gives_chunks_back_instead_of_holding_eight(monkeypatch):
    """with room for little more than the five fields themselves the trial keeps five chunks and re-allocates the slowest"""
    import torch
    off, q0 = _run(monkeypatch, "0")
    free, total = torch.cuda.mem_get_info(0)
    chunk = 36 * 5400 * 16 * 72 * 8                      # a scratch field of ne30/q35: qsize + 1 planes
    hog = torch.empty(free - (24 << 30) - 6 * chunk, dtype=torch.uint8, device="cuda:0")   # leaves 24 GiB (the reserve) + 6 chunks
    try:
        monkeypatch.setenv("TSE_PLACEMENT_GOOD", "1e9")
        on, q1 = _run(monkeypatch, "12")
    finally:
        del hog
        torch.cuda.empty_cache()
    assert on["tried"] == 12 and len(set(on["chosen"])) == 5, on
    assert np.array_equal(q0.view(np.uint64), q1.view(np.uint64))
