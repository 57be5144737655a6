"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/ocrgpu.h declares,
and refuses to work without a GPU (no silent CPU fallback)."""
import os
import re

import pytest

from ocr_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_agree():
    hdr = open(os.path.join(ROOT, "include", "ocrgpu.h")).read()
    declared = set(re.findall(r"\b(ocrgpu_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"ocrgpu_engine", "ocrgpu_engine_t", "ocrgpu_reclaim_t", "ocrgpu_digest_term"}   # (types; a static inline of the header)
    assert declared == set(engine.EXPORTS)
    lib = engine.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ocrgpu_abi_version() == 6


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.OcrGpuError) as ei:
        engine.Engine(16)
    assert ei.value.code == -19  # OCRGPU_ENODEV
