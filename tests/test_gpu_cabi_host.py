"""A plain C program drives libocrgpu (tests/cabi/two_partitions.c, built by __graft_entry__.build): two engines in one process
joined by peer windows, a chain of EDTs alternating between them, the worker loop across partitions -- and the same chain on one
partition, which must give the same levels and HASH values.  No Python between the program and the library."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_two_partitions_over_windows():
    exe = os.path.join(ROOT, "ocr_amd", "bin", "cabi_two_partitions")
    assert os.path.exists(exe), "run __graft_entry__.build() first (programs are built on the CPU, not inside a GPU test)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok: 64-link chain"), (r.stdout[-500:], r.stderr[-1500:])
