"""bench.py's host-side helpers (no GPU): the per-rank motion of the multi-GPU workload and the readers of the committed
rocprofv3 counter summaries that fill `roofline.traffic` and `roofline.executed`."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_rank_moves_inside_the_search_range():
    """A rank whose frames move by more than 16 output pixels has no match anywhere: its motion stage would search in
    full (ten times slower) and, timed by the slowest rank, wreck the multi-GPU figure."""
    b = _bench()
    seen = set()
    for rank in range(64):
        dx, dy = b.rank_motion(rank)
        assert abs(2 * dx) <= 16 and abs(2 * dy) <= 16, (rank, dx, dy)
        if rank < 8:
            seen.add((dx, dy))
    assert b.rank_motion(0) == (3, -2)                       # the single-GPU workload of BASELINE config 3
    assert len(seen) == 8                                    # eight GPUs, eight different motions


def test_counter_summaries_are_readable():
    """The committed profiles feed two fields of the bench line; a format drift must not turn them into None unnoticed."""
    b = _bench()
    traffic, src = b.pmc_traffic(["lfg::motion_tiled", "lfg::motion_prefilter", "lfg::motion_resolve", "lfg::motion_hint", "lfg::motion_order"])
    assert src and src.startswith("profiles/") and traffic and 5e7 < traffic < 5e9
    ex = b.pmc_executed("lfg::motion_prefilter")
    assert ex is not None and ex["launches_per_call"] == b.PREFILTER_LAUNCHES_PER_CALL
    assert 0.05 < ex["valu_issue_utilisation"] < 1.0 and ex["valu_wave_instructions"] > 1e7
    assert b.pmc_traffic("lfg::scale_2x")[0] > 1e6
