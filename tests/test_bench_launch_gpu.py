"""bench.py's N > 1 launch path on a 1-GPU box: `python bench.py --gpus 2 ...` outside torchrun re-launches itself under
torch.distributed.run (relaunch_under_torchrun), two ranks run the data-parallel step - packed all-gather, index-then-fetch negatives, the gradient
reducer - and rank 0 prints ONE JSON line carrying `comm.per_rank_*`; rank 1 leaves through the rank != 0 exit path.  Both ranks sit on device 0
and the process group is gloo (MICO_BENCH_ONE_DEVICE / MICO_BENCH_BACKEND: a 1-GPU box cannot host two RCCL ranks) - everything of the launch
shape the driver's SCALE run uses (the reference's: data/scripts/run_vision_captioner.sh:1-12, one process per GPU) except RCCL itself."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("backward", ["staged", "direct"])
def test_bench_two_ranks_on_one_device(cuda, backward):
    env = dict(os.environ, MICO_BENCH_ONE_DEVICE="1", MICO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    # (the staged run keeps the extras rank 0 runs at N > 1 - the in-run parity measurement, alone, while rank 1 waits at the final barrier: its
    # alignment steps must stay out of the collectives, distributed.local_only)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--batch", "4", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + (["--direct-backward", "--no-extras"] if backward == "direct" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines            # exactly one line on stdout, from rank 0
    d = json.loads(lines[0])
    assert len(lines[0]) < 8192              # the whole line fits a driver's tail
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert all(math.isfinite(v) for v in d["losses"].values()) and set(d["losses"]) == {"loss_itc", "loss_itm", "loss_cap"}
    c = d["comm"]
    assert c["world_size"] == 2 and len(c["per_rank_samples_per_s"]) == 2 and len(c["per_rank_exposed_reduce_ms"]) == 2
    assert c["backend"].startswith("gloo") and c["rccl_ranks"] == 0
    assert d["roofline"] is not None and "step_frac" in d["roofline"]
    assert ("staged" in d["config"]["backward"]) == (backward == "staged")
    if backward == "staged":
        assert d["parity"]["worst"] < 1e-3 and d["parity"]["tensors"] >= 12, d["parity"]      # (depth-2 goldens; the full-depth one needs the full model)
        assert "secondary" not in d and "cpu_baseline" not in d                                 # N = 1 objects
