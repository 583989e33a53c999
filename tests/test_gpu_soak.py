"""A short run of tools/soak.py in the GPU suite: threads firing random transcribe / submit / batch / streaming requests at a two-replica
ASRModel; every transcript must equal the one the same request gave alone (batch invariance, bit for bit), no exceptions."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode,sched", [("native", "batch"), ("int8", "batch"), ("native", "continuous"), ("int8", "continuous")])
def test_randomised_concurrency_soak(mode, sched):
    """two replicas x two slots each; batch: every slot runs whole batches; continuous: rows join and leave a running greedy loop"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "6", "6", mode, "tiny", sched], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "errors: 0" in r.stdout


@pytest.mark.parametrize("mode", ["native", "int8"])
def test_no_batch_composition_changes_a_result_full_dimensions(mode):
    """tools/find_batch_dependence.py at full model dimensions: random batches of 2-6 requests (0.1-40 s, one or two windows, mixed
    budgets) against each request's solo transcript.  This is the search that found the int8 GEMM's ragged-tail rows reading the
    quantisation data of the batch's first rows (the last request of some compositions differed)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "find_batch_dependence.py"), mode, "60"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 differing batches" in r.stdout
