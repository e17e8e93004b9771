"""include/putslam_shard.h: sharding over the GPUs of a node for C / C++ hosts (RCCL underneath), exercised with a world of
one on the one-GPU box through demos/cpp/demo_sequences_multi_gpu -- a process without torch.  The exchanges are bench.py's:
parameter block from rank 0 (ncclBroadcast), 72-byte per-pair records to rank 0 (ncclGather), where the reference's only
sequential step composes the trajectory (reference src/PUTSLAM/PUTSLAM.cpp:735-740)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from putslam_amd import synth
from putslam_amd._abi import EST_FIXED, EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "demos", "cpp", "demo_sequences_multi_gpu")


def _exe():
    if not os.path.exists(EXE):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build_dropin()
    return EXE


def _write_sequence(path, seq):
    F, cap = seq["desc"].shape[:2]
    with open(path, "wb") as f:
        np.array([F, cap], np.int32).tofile(f)
        np.ascontiguousarray(seq["nkpts"], np.int32).tofile(f)
        np.ascontiguousarray(seq["desc"], np.uint8).tofile(f)
        np.ascontiguousarray(seq["pts"], np.float32).tofile(f)


CASES = [("one-process", "queue+async", "fixed", EST_FIXED, 768, 1, 14), ("rank-per-process", "queue+async", "fixed", EST_FIXED, 512, 1, 31),
         ("one-process", "queue+async+thread", "fixed", EST_FIXED, 512, 1, 31), ("rank-per-process", "queue+async+thread", "ransac", EST_RANSAC, 487, 0, 14),
         ("one-process", "blocking", "fixed", EST_FIXED, 768, 1, 14), ("rank-per-process", "blocking", "ransac", EST_RANSAC, 487, 0, 14),
         ("one-process", "queue+async", "ransac", EST_RANSAC, 487, 0, 31)]


@pytest.mark.parametrize("launch,mode,estimator,est,H,ev,frames", CASES)
def test_native_gather_equals_python_batch_and_oracle(ctx, oracle, tmp_path, launch, mode, estimator, est, H, ev, frames):
    """The records rank 0 gathers natively are the bytes sharding.pack_records makes of the oracle's results for the same
    sequence and seed -- through the default host loop (ps_shard_submit_all: a PsBatchQueue per member, records packed on the
    chains; ps_shard_gather_records_async / ps_shard_wait: the gather on the communication stream, read one step later), the
    same with the member on a host thread of its own (what a process that drives several GPUs runs; PUTSLAM_SHARD_THREADS=1
    forces it for one member), and rounds 1 - 5's loop (ps_vo_pairs_device on the member's context + the blocking gather).
    31 frames = 30 pairs: the queue splits the batch 13 + 17 over its two chains; 13 pairs go to the chains in turn."""
    from putslam_amd import sharding
    seq = synth.make_sequence(frames, 700, config=3, index=4242)
    _write_sequence(tmp_path / "seq0.bin", seq)
    seed = 0xB0B0
    dump = tmp_path / "records.bin"
    cmd = [_exe(), "--sequence-prefix", str(tmp_path / "seq"), "--estimator", estimator, "--hyp", str(H), "--error-version", str(ev),
           "--seed", str(seed), "--steps", "7", "--dump", str(dump), "--traj-prefix", str(tmp_path / "traj")]
    cmd += ["--gpus", "1"] if launch == "one-process" else ["--rank", "0", "--world", "1", "--id-file", str(tmp_path / "id.bin")]
    if mode == "blocking":
        cmd.append("--blocking")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode.endswith("thread"):
        env["PUTSLAM_SHARD_THREADS"] = "1"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert ("blocking gather" if mode == "blocking" else "asynchronous gather") in p.stdout
    got = np.fromfile(dump, np.float32).reshape(1, frames - 1, sharding.RECORD_FLOATS)
    prm = default_ransac_params(ev)
    cfg, _ = make_config(est, H, seed=seed)                      # rank 0: seed + 0
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"], threads=4)
    want = sharding.pack_records(c["pose"], c["stats"]["numInliers"], c["stats"]["numMatchesIn"]).numpy()
    assert got[0].tobytes() == want.tobytes()
    # the trajectory rank 0 composed from them (VOTrajectory, PUTSLAM.cpp:735-740,1006-1016) is sharding.compose_trajectory's
    inc = want[:, :16].reshape(-1, 4, 4).transpose(0, 2, 1)
    traj = sharding.compose_trajectory(inc)
    lines = open(tmp_path / "traj0.txt").read().strip().split("\n")
    assert len(lines) == frames
    for k, line in enumerate(lines):
        assert line.split()[1:] == sharding.format_tum_line(0.0, traj[k]).split()[1:], k


def test_native_demo_on_synthetic_sequences(tmp_path):
    """The demo's own frames (known motion): exit code 0 = every rank's block arrived and every increment is within 5e-3 of
    the ground truth."""
    p = subprocess.run([_exe(), "--gpus", "1", "--frames", "30", "--kpts", "1200", "--estimator", "ransac", "--hyp", "487",
                        "--error-version", "0", "--steps", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "29 of 29 increments accepted" in p.stdout and "records gathered over RCCL" in p.stdout


@pytest.mark.parametrize("outstanding,steps,threaded", [(99, 12, False), (99, 12, True), (1, 5, False), (7, 11, False)])
def test_native_gathers_the_host_never_waits_for_are_issued_all_the_same(tmp_path, outstanding, steps, threaded):
    """The gather of a step is issued once the host has seen its records packed (ps_shard.hip: member_flush) -- at a later submit /
    gather call, or in ps_shard_wait.  A host that keeps more gathers outstanding than there are record blocks (--outstanding 99:
    it never waits before the end) has them issued when their block is needed again (slot_retire), tickets in order; one that waits
    at once (--outstanding 1) has them issued in the wait.  The last step's records are those of the default loop, byte for byte."""
    seq = synth.make_sequence(14, 600, config=3, index=777)
    _write_sequence(tmp_path / "seq0.bin", seq)
    dumps = []
    for tag, extra in (("default", []), ("probe", ["--outstanding", str(outstanding)])):
        dump = tmp_path / (tag + ".bin")
        cmd = [_exe(), "--gpus", "1", "--sequence-prefix", str(tmp_path / "seq"), "--estimator", "fixed", "--hyp", "512", "--error-version", "1",
               "--seed", "4321", "--steps", str(steps), "--dump", str(dump)] + extra
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if threaded and tag == "probe":
            env["PUTSLAM_SHARD_THREADS"] = "1"
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        dumps.append(np.fromfile(dump, np.float32))
    assert dumps[0].size == 13 * 18 and dumps[0].tobytes() == dumps[1].tobytes()
    assert np.abs(dumps[0].reshape(13, 18)[:, :16]).sum() > 0
