"""A slice of the randomised soak (tests/fuzz_gpu.py) collected under `-m gpu`: 300 random configurations of size,
inlier ratio, noise, error mode, estimator, thresholds, camera scale and seed, HIP path vs oracle, every output
field compared (match list, mask, pose bytes, all statistics)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 2026])
def test_fuzz_slice(ctx, oracle, seed):
    import fuzz_gpu
    assert fuzz_gpu.run(150, seed, max_kpts=1200, ctx=ctx, verbose=False) == 0


def test_fuzz_batch_slice(oracle):
    """Random batches through the staged scoring: staged == complete on every pair, == oracle on sampled pairs."""
    import fuzz_gpu
    assert fuzz_gpu.run_batch(12, 77, verbose=False) == 0
