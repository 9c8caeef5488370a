"""Host-side helpers of bench.py that need no GPU: the launcher's share of host threads per rank, and the per-width table
built from tfa_profile_launches records."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_host_threads_are_shared_out_over_the_ranks(monkeypatch):
    monkeypatch.setenv("OMP_NUM_THREADS", "16")
    assert [bench.host_threads_per_rank(n) for n in (1, 2, 4, 8, 32)] == [16, 8, 4, 2, 1]
    monkeypatch.delenv("OMP_NUM_THREADS")
    assert bench.host_threads_per_rank(1) >= 1 and bench.host_threads_per_rank(10 ** 6) == 1


def test_width_table_groups_bootstrap_launches():
    from tfhe_aes_amd.engine import LAUNCH_DTYPE
    rec = np.array([(128, 0, 1, 2.0), (128, 1, 0, 0.05), (480, 0, 2, 4.0), (512, 0, 2, 4.0), (3984, 0, 8 | (2 << 8), 20.0), (70000, 0, 8, 350.0)], LAUNCH_DTYPE)
    t = bench.width_table(rec)
    rows = {r["width_le"]: r for r in t["bootstrap_launches_by_width"]}
    assert rows[128]["launches"] == 1 and rows[128]["kpbs_per_s"] == 64.0
    assert rows[512]["launches"] == 2 and rows[512]["bootstraps"] == 992 and rows[512]["kpbs_per_s"] == 124.0
    assert rows[4096]["bootstraps"] == 3984 and rows[None]["bootstraps"] == 70000
    assert 256 not in rows and t["keyswitch_ms"] == 0.05 and t["bootstrap_ms"] == 380.0
