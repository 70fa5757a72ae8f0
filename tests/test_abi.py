"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/monortm_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

from common import ROOT
from monortm_amd import _build, api


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "monortm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(monortm_hip_\w+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(api.SYMBOLS)


def test_library_builds_and_exports_every_symbol():
    so = _build.build_hip()
    lib = ctypes.CDLL(so)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/monortm_hip.h but not exported by {so}"
    api.load_library()


def test_no_gpu_means_loud_failure(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from monortm_amd import synth, tape3

    t3 = str(tmp_path / "TAPE3")
    tape3.write_tape3(t3, synth.synthetic_lines(20))
    with pytest.raises(api.MonoRTMError):
        api.MonoRTM(t3, 0.3, 30.0)


def test_product_does_not_import_oracle():
    """The shipped package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "monortm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".f90", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in src and "liboracle" not in src and "monortm_oracle" not in src, f
