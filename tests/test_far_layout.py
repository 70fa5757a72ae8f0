"""far_kernel's workgroup placement and interval numbering (device_common.hpp: far_xcd_share, far_xcd_items, far_level_*) checked
on the host: hipcc compiles the header's host side, no GPU is touched.  A (interval, molecule) that no workgroup serves would be a far
field silently missing from the sums; one served twice would count twice."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (shutil.which(HIPCC) or os.path.exists(HIPCC)), reason="hipcc not found")
def test_far_placement_serves_every_interval_and_molecule_once(tmp_path):
    exe = tmp_path / "far_layout_check"
    src = os.path.join(ROOT, "tests", "native", "far_layout_check.cpp")
    subprocess.run([HIPCC, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src, "-o", str(exe)], check=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout
    assert out.startswith("ok "), out
