"""bench.py's N > 1 branch (parameter broadcast from rank 0, per-chain asynchronous gather of the 72-byte per-pair
records to rank 0, closing fence) exercised on the GPU box: two ranks as two fresh child processes sharing the one
GPU, collectives over gloo (PUTSLAM_BENCH_BACKEND, a test hook: the measured configuration is RCCL, one rank per GPU).
Rank 0's gathered records must equal what two independent single-rank runs produce for the same sequences."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--frames", "14", "--kpts", "700", "--hyp", "768", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in kw.items()})
    return e


@pytest.mark.parametrize("streams", [1, 3])
def test_bench_two_ranks_on_one_gpu_equal_two_single_rank_runs(tmp_path, streams):
    bench = os.path.join(ROOT, "bench.py")
    port = _free_port()
    multi = tmp_path / "multi.npy"
    procs = []
    for r in range(2):
        env = _env(RANK=r, LOCAL_RANK=r, WORLD_SIZE=2, MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   PUTSLAM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=0)
        cmd = [sys.executable, bench, "--gpus", "2", "--streams", str(streams), "--dump-records", str(multi)] + ARGS
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    import json
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["pairs_per_step"] == 26 and j["value"] > 0
    got = np.load(multi)
    assert got.shape == (2, 13, 18)
    for r in range(2):
        single = tmp_path / f"single{r}.npy"
        p = subprocess.run([sys.executable, bench, "--gpus", "1", "--streams", str(streams), "--as-rank", str(r),
                            "--dump-records", str(single)] + ARGS, env=_env(WORLD_SIZE=1, RANK=0, LOCAL_RANK=0),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        one = np.load(single)
        assert one.shape == (1, 13, 18)
        assert got[r].tobytes() == one[0].tobytes(), f"rank {r}: gathered records differ from the single-rank run"
    assert got[0].tobytes() != got[1].tobytes()      # the two ranks worked on different sequences
