"""The C-ABI boundary used from plain C: tests/abi/c_client.c (built by __graft_entry__.build()) allocates
with the HIP runtime API, calls libdsic_hip.so through include/dsic_hip.h and checks the results on the host."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "abi", "_bin", "c_client")


@pytest.mark.gpu
def test_plain_c_client_of_the_library():
    if not os.path.exists(BIN):
        pytest.fail(f"{BIN} is missing: run __graft_entry__.build()")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_client ok" in r.stdout, (r.stdout, r.stderr)
