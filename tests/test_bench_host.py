"""Host-side pieces of bench.py that need no GPU: the path's roofline model, the synthetic prompt, argument handling of
the self-spawning launcher (the parent must fail cleanly, before touching any GPU, when a rank fails)."""
import os
import subprocess
import sys

import pytest

import bench
from whisper_amd.config import dims_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_path_roofline_model_for_the_headline_config():
    # whisper-small, 128 windows, 170-token prompt, 128 generated tokens, bf16: the figure path_frac is quoted against
    ms = bench.path_roofline_ms(dims_for("small"), 128, 170, 128, 2)
    assert 190.0 < ms < 200.0, ms
    # decode bytes dominate and scale with the batch; the dense part scales with it too
    assert bench.path_roofline_ms(dims_for("small"), 64, 170, 128, 2) < 0.55 * ms
    # fp32 streams twice the bytes
    assert bench.path_roofline_ms(dims_for("small"), 128, 170, 128, 4) > 1.7 * ms


def test_prompt_shape_and_determinism():
    p = bench.make_prompt(0)
    assert len(p) == 170 and p[0] == bench.SOT_PREV and p[161:165] == [bench.SOT, bench.LANG_EN, bench.TRANSCRIBE, bench.TS_BEGIN]
    assert p == bench.make_prompt(5)                      # the same prompt text for every window
    assert all(1000 <= t < 40000 for t in p[1:161] + p[165:])


def test_world_size_mismatch_is_rejected_before_any_gpu_work():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
