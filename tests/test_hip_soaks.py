"""GPU: a pinned slice of the round-3 soak tools as regression tests (each runs as a child process: one more process on
the card at a time).  tools/soak_live_rows.py -- the row lists of the grad step against the dense step over random shapes
(every attention kernel family, +- attention tower / id table / input gradient, NaN-filled free memory);
tools/soak_compact.py -- the device-compacted encoder against the padded pipeline (bitwise for prefix masks)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("script,count,seed0", [("soak_live_rows.py", 16, 0), ("soak_compact.py", 24, 0)])
def test_soak_slice(script, count, seed0):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(count), str(seed0)], capture_output=True,
                       text=True, timeout=550, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0, tail
    assert f"{count} configurations, 0 failures" in r.stdout, tail
