"""A short run of tools/soak.py in the GPU suite: threads firing random transcribe / submit / batch / streaming requests at a two-replica
ASRModel; every transcript must equal the one the same request gave alone (batch invariance, bit for bit), no exceptions."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["native", "int8"])
def test_randomised_concurrency_soak(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "6", "6", mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "errors: 0" in r.stdout
