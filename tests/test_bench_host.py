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


def test_a_failing_rank_stops_the_others_and_sets_the_exit_code(tmp_path):
    """spawn_ranks: rank 1 exits with code 3 at start-up while rank 0 would wait a minute (as it would in the
    rendezvous); the launcher must stop rank 0 and report 3 within seconds."""
    import time
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "time.sleep(60)\n")
    t0 = time.time()
    rc = bench.spawn_ranks(2, [], script=str(script))
    assert rc == 3 and time.time() - t0 < 20


def test_ranks_that_finish_cleanly_give_zero(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text("import os\nassert int(os.environ['LOCAL_RANK']) == int(os.environ['RANK'])\n")
    assert bench.spawn_ranks(3, [], script=str(script)) == 0
