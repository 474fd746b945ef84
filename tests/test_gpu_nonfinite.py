"""NaN / Inf samples never fault the device and never stick: tools/nonfinite_probe.py in a process of its own (a device fault ends the
process that caused it -- it must fail this test, not the suite)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_non_finite_samples_never_fault_and_never_stick(gpu):
    """every entry point (offline renders on ten plans incl. both RSNT forms, with line results and a carried state; the spectrum stream;
    the Oscilloscope in four trigger / interpolation modes; the Vectorscope in its three envelope modes) takes blocks that hold NaN and
    +Inf: every call returns, frames in front of the first bad sample are bit-identical to the render with the bad samples zeroed, and
    clean audio afterwards (same plan; a new handle) gives the clean result bit for bit.  The full list (-Inf, 3e38): the tool without
    `quick`, profiles/r06d/nonfinite_probe.txt"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nonfinite_probe.py"), "quick"], capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    assert "problems: 0" in r.stdout, tail
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 40, tail
