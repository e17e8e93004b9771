"""First contact with more than one physical GPU (BASELINE configs[3]: sequences sharded one per GPU, RCCL gather of poses).
Every test here ACTIVATES ITSELF when the box exposes two devices or more and is skipped -- with the device count in the reason
-- on the one-GPU boxes of this pool, where N > 1 RCCL ranks cannot exist (the one-GPU forms of the same control flow:
tests/test_gpu_multirank.py over gloo, tests/test_gpu_shard_native.py with a world of one).  The one sequential step the gather
feeds: reference src/PUTSLAM/PUTSLAM.cpp:735-740."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from putslam_amd import synth
from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NDEV = torch.cuda.device_count()          # (does not initialise the GPU)
needs_two = pytest.mark.skipif(NDEV < 2, reason=f"needs two GPUs or more for real RCCL ranks; this box exposes {NDEV}")
EXE = os.path.join(ROOT, "demos", "cpp", "demo_sequences_multi_gpu")
ARGS = ["--frames", "26", "--kpts", "700", "--hyp", "768", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--repeats", "2",
        "--no-other-modes"]


def _env(**kw):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "PUTSLAM_BENCH_BACKEND"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in kw.items()})
    return e


def _write_sequence(path, seq):
    F, cap = seq["desc"].shape[:2]
    with open(path, "wb") as f:
        np.array([F, cap], np.int32).tofile(f)
        np.ascontiguousarray(seq["nkpts"], np.int32).tofile(f)
        np.ascontiguousarray(seq["desc"], np.uint8).tofile(f)
        np.ascontiguousarray(seq["pts"], np.float32).tofile(f)


def test_device_count_is_visible():
    """Not skipped anywhere: the count the self-activating tests key on, in the log of every GPU run."""
    print(f"visible HIP devices: {NDEV}")
    assert NDEV >= 1


@needs_two
def test_bench_two_gpus_over_rccl_equals_two_single_rank_runs(tmp_path):
    """`python bench.py --gpus 2` started plainly (the launcher spawns one rank per GPU, backend nccl = RCCL, records gathered
    through the library's batch-queue chains) == two independent `--as-rank r` runs, byte for byte."""
    bench = os.path.join(ROOT, "bench.py")
    multi = tmp_path / "multi.npy"
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--dump-records", str(multi)] + ARGS, env=_env(), capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["backend"] == "nccl" and j["value"] > 0
    assert j["config"]["pairs_per_step"] == 50 and j["scaling"] == "weak"
    got = np.load(multi)
    assert got.shape == (2, 25, 18)
    for r in range(2):
        single = tmp_path / f"single{r}.npy"
        q = subprocess.run([sys.executable, bench, "--gpus", "1", "--as-rank", str(r), "--dump-records", str(single)] + ARGS,
                           env=_env(), capture_output=True, text=True, timeout=900)
        assert q.returncode == 0, q.stdout + q.stderr
        assert got[r].tobytes() == np.load(single)[0].tobytes(), f"rank {r}"
    assert got[0].tobytes() != got[1].tobytes()


def _oracle_records(oracle, seqs, est, H, ev, seed):
    from putslam_amd import sharding
    out = []
    for r, seq in enumerate(seqs):
        cfg, _ = make_config(est, H, seed=seed + r)              # one sequence per GPU: rank r draws from seed + r
        c = oracle.vo_pairs(default_ransac_params(ev), cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"], threads=4)
        out.append(sharding.pack_records(c["pose"], c["stats"]["numInliers"], c["stats"]["numMatchesIn"]).numpy())
    return np.stack(out)


@needs_two
@pytest.mark.parametrize("launch", ["one-process", "rank-per-process"])
@pytest.mark.parametrize("mode", ["queue+async", "blocking"])
def test_native_two_gpus_equal_the_oracle(oracle, tmp_path, launch, mode):
    """include/putslam_shard.h on two GPUs, both launch shapes (ncclCommInitAll with a host thread per member /
    ncclCommInitRank with one process per GPU), both host loops: rank 0's gathered records are the bytes
    sharding.pack_records makes of the oracle's results for each rank's sequence and seed."""
    seqs = [synth.make_sequence(26, 700, config=3, index=7000 + r) for r in range(2)]
    for r, s in enumerate(seqs):
        _write_sequence(tmp_path / f"seq{r}.bin", s)
    seed, H, ev = 0xB0B0, 768, 1
    dump = tmp_path / "records.bin"
    base = [EXE, "--sequence-prefix", str(tmp_path / "seq"), "--estimator", "fixed", "--hyp", str(H), "--error-version", str(ev),
            "--seed", str(seed), "--steps", "3", "--dump", str(dump)] + (["--blocking"] if mode == "blocking" else [])
    if launch == "one-process":
        p = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=600, env=_env())
        assert p.returncode == 0, p.stdout + p.stderr
    else:
        procs = [subprocess.Popen(base + ["--rank", str(r), "--world", "2", "--id-file", str(tmp_path / "id.bin")], env=_env(),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = [q.communicate(timeout=600) for q in procs]
        for q, (so, se) in zip(procs, outs):
            assert q.returncode == 0, so + se
    got = np.fromfile(dump, np.float32).reshape(2, 25, 18)
    want = _oracle_records(oracle, seqs, EST_FIXED, H, ev, seed)
    assert got.tobytes() == want.tobytes()


@needs_two
def test_native_all_gpus_synthetic_sequences():
    """Every visible GPU, the demo's own frames (known motion): exit code 0 = every rank's block arrived on rank 0 and every
    increment is within 5e-3 of the ground truth."""
    p = subprocess.run([EXE, "--frames", "30", "--kpts", "1200", "--estimator", "ransac", "--hyp", "487", "--error-version", "0",
                        "--steps", "3"], capture_output=True, text=True, timeout=900, env=_env())
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("29 of 29 increments accepted") == NDEV
