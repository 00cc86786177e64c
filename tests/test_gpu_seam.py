"""SURVEY.md 8 row b4 as code: the StageBuildGraph binding of INTEGRATION.md section 1 -- oracle/ref/seam_driver.cc, compiled against
the REFERENCE's headers and linked with the reference's objects by oracle/ref/build_ref.sh -- run on the GPU: the reference's own
types go in (vecbvec, VecPQVec, vec<int32_t>), libsnk does count + unitigs through the C ABI, the reference's buildHBVFromEdges
finishes; unitigs and graph equal the reference's own run (golden dumps of snref_driver)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import goldens
import refio

pytestmark = pytest.mark.gpu

SEAM = refio.REF_DRIVER.parent / "snref_seam"


@pytest.mark.parametrize("name", ["adversarial", "synth_20k_err", "synth_2k_err"])
def test_df_seam_stub_against_reference_headers(snk, tmp_path, name):
    from supernova_amd import synth
    if not SEAM.exists():
        pytest.skip("oracle/_ref/snref_seam not built (needs /root/reference in the build container)")
    c = goldens.load(name)
    asc = synth.codes_to_ascii(c.codes)          # (an N of the original reads is an A on both sides: ParseBarcodedFastqs.cc:87-88)
    refio.write_snkrd(tmp_path / "in.snkrd", c.lens, asc, c.quals, c.bc, c.ign_bc_below)
    r = subprocess.run([str(SEAM), str(tmp_path / "in.snkrd"), str(tmp_path / "out")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SNREF_SEAM" in r.stdout
    assert (tmp_path / "out" / "unitigs.txt").read_text().split() == c.exp_unitigs
    assert (tmp_path / "out" / "hbv.txt").read_text() == c.exp_hbv
