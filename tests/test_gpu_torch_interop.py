"""The way bench.py uses the library -- torch-owned device buffers and streams handed to the
device-pointer entry points -- checked against the host-pointer forms.  Runs in its own process
(torch has to be imported before libgwscan.so pulls in ROCm's HIP runtime)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_torch_buffers_and_streams():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "torch_interop_script.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["device_pointer_forms_match_host_forms"]
    assert out["counts_only_leaves_stats_untouched"]
    assert out["upload_download_round_trip"]
    assert out["upload_stream_rows_GBps"] > 0.5
