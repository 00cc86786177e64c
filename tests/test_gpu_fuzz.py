"""A short fixed-seed slice of the randomised parity sweep (tests/tools/fuzz_parity.py): random genomes with planted repeats,
tandem runs and palindromes, random K / read length / filters / bucket counts / partition capacity, through the one-GPU path
(both graph stages, device HBV, per-group runs) and 2-8 simulated ranks, everything against the C oracle."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("seed", [7, 8])
def test_fuzz_slice(snk, seed):
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "fuzz_parity.py"), "12", str(seed)], capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert "12 of 12 cases bit-exact" in r.stdout, tail
