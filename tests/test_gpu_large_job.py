"""The large-job tool that bench.py's config.large_job rows run (tools/r6_full_job.py), at fixture size."""
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("mode", ["plain", "per_barcode_graphs", "k60"])
def test_the_large_job_tool_at_fixture_size(mode):
    """tools/r6_full_job.py -- what bench.py's config.large_job rows run at 800 M / 1.2 B reads -- at 2 M reads on a context that is told it
    has 256 MB, so that the same bucket-range passes run: its checks (every count >= min_freq, spectrum and unitig lengths add up to the table,
    the second call equal to the first) hold and it says so in its rows."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, SNK_TUNING="plan_mem_mb=256")
    if mode == "per_barcode_graphs":
        env["GROUPED"] = "1"
    if mode == "k60":
        env["K"] = "60"
    pr = subprocess.run([sys.executable, str(ROOT / "tools" / "r6_full_job.py"), "2e6", "5e5", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    rows = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 2 and all(r["min_count"] >= 3 and r["spectrum_adds_up"] and r["unitig_lengths_add_up"] and r["same_as_first_call"] for r in rows)
    assert rows[1]["passes"] > 1 and rows[1]["retained_kmers"] > 100_000
