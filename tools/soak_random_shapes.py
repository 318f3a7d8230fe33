#!/usr/bin/env python
"""One-off soak: the seeded random-shape parity tests (tests/test_hip_random_shapes.py) over seeds beyond the ones the
test-suite pins -- forward and gradients vs the CPU oracle."""
import sys
import os
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_hip_random_shapes as T  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    for fn in (T.test_random_bi_encoder_matches_oracle, T.test_random_bi_encoder_gradients_match_oracle):
        try:
            fn(seed)
        except Exception as e:  # noqa: BLE001
            bad.append((seed, fn.__name__, repr(e)[:200]))
            traceback.print_exc(limit=1)
print(f"seeds {lo}..{hi - 1}: {len(bad)} failures")
for b in bad:
    print(b)
sys.exit(1 if bad else 0)
