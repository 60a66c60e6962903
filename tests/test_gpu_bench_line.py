"""bench.py's line under --force-comm (the N > 1 code path on one GPU: a one-rank nccl group, the library's
RCCL communicator in the step): the fields that make an N > 1 line self-verifying -- how many ranks RCCL says
the communicator spans, what every rank held, that the value comes from the RCCL path -- are present and right."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_force_comm_line_reports_rccl_ranks_and_per_rank_shares():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--views", "20000", "--force-comm", "--no-solves", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["rccl_nranks"] == 1 and "rehearsal" not in out and out["value"] > 0
    assert out["per_rank"] == [dict(rank=0, views=20000, obs=20000 * 88, kernel_region_ms=out["per_rank"][0]["kernel_region_ms"])]
    assert out["per_rank"][0]["kernel_region_ms"] > 0
    assert out["strong_scaling"]["rccl_nranks"] == 1 and out["strong_scaling"]["per_rank"][0]["views"] == 20000
    assert "RCCL" in out["config"]["collective"]
    for k in ("roofline", "roofline_hbm"):
        assert k in out
