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
ARGS = ["--frames", "14", "--kpts", "700", "--hyp", "768", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
        "--repeats", "2", "--no-other-modes"]
TINY = ["--kpts", "300", "--hyp", "256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--repeats", "1",
        "--no-other-modes", "--streams", "1"]


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
    assert j["config"]["world_size"] == 2 and j["config"]["backend"] == "gloo"     # the line shows what the ranks ran on
    assert j["repeats"] == 2 and j["value_min"] <= j["value"] <= j["value_max"]
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


def _launch_ranks(world, extra, dump):
    bench = os.path.join(ROOT, "bench.py")
    port = _free_port()
    procs = []
    for r in range(world):
        env = _env(RANK=r, LOCAL_RANK=r, WORLD_SIZE=world, MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   PUTSLAM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=0)
        cmd = [sys.executable, bench, "--gpus", str(world), "--dump-records", str(dump)] + extra
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    import json
    return json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """World size 8 (BASELINE configs[3]'s shape: one sequence per rank, gather of 72 B per pair to rank 0), tiny sizes,
    eight processes sharing the one GPU over gloo: every rank's block arrives on rank 0 and equals that rank's own
    single-process run."""
    extra = ["--frames", "5"] + TINY
    multi = tmp_path / "multi8.npy"
    j = _launch_ranks(8, extra, multi)
    assert j["n_gpus"] == 8 and j["config"]["world_size"] == 8 and j["config"]["pairs_per_step"] == 32
    assert j["scaling"] == "weak"
    got = np.load(multi)
    assert got.shape == (8, 4, 18)
    assert len({got[r].tobytes() for r in range(8)}) == 8                  # eight different sequences
    bench = os.path.join(ROOT, "bench.py")
    for r in (0, 3, 7):
        single = tmp_path / f"s{r}.npy"
        p = subprocess.run([sys.executable, bench, "--gpus", "1", "--as-rank", str(r), "--dump-records", str(single)] + extra,
                           env=_env(WORLD_SIZE=1, RANK=0, LOCAL_RANK=0), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        assert got[r].tobytes() == np.load(single)[0].tobytes(), f"rank {r}"


def test_bench_one_sequence_sharded_eight_ways_equals_unsharded(tmp_path):
    """--shard sequence: ONE sequence split over eight ranks with a one-frame halo (sharding.shard_sequence); the pair
    records rank 0 gathers, put back in order, are byte-identical to the unsharded single-rank run (pair p keeps its
    hypothesis stream seed + p wherever it runs)."""
    from putslam_amd import sharding
    frames = 20
    extra = ["--frames", str(frames), "--shard", "sequence"] + TINY
    multi = tmp_path / "shard8.npy"
    j = _launch_ranks(8, extra, multi)
    assert j["scaling"] == "strong" and j["config"]["pairs_per_step"] == frames - 1 and j["config"]["shard"] == "sequence"
    got = np.load(multi)                                              # (8, largest shard, 18), short shards zero-padded
    assert got.shape == (8, 3, 18)
    bench = os.path.join(ROOT, "bench.py")
    single = tmp_path / "unsharded.npy"
    p = subprocess.run([sys.executable, bench, "--gpus", "1", "--dump-records", str(single), "--frames", str(frames)] + TINY,
                       env=_env(WORLD_SIZE=1, RANK=0, LOCAL_RANK=0), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    want = np.load(single)[0]
    assert want.shape == (frames - 1, 18)
    back = []
    for r in range(8):
        sh = sharding.shard_sequence(frames, 8, r)
        n = sh["pair_hi"] - sh["pair_lo"]
        back.append(got[r][:n])
        assert not got[r][n:].any()                                   # padding only
    back = np.concatenate(back)
    assert back.tobytes() == want.tobytes()
    # and the trajectory rank 0 would compose from them (PUTSLAM.cpp:735-740) is the unsharded one
    inc = back[:, :16].reshape(-1, 4, 4).transpose(0, 2, 1)
    assert sharding.compose_trajectory(inc).tobytes() == \
        sharding.compose_trajectory(want[:, :16].reshape(-1, 4, 4).transpose(0, 2, 1)).tobytes()


def test_bench_rccl_branch_through_the_batch_queue(tmp_path):
    """The default submission of a multi-rank run: ONE ps_batch_queue_submit per step (the library's two chains, 45 % / 55 %),
    records packed and gathered per chain on the chains' own streams (torch sees them as ExternalStreams) -- with a world of
    one over RCCL, 25 pairs (13 would go to the queue's chains in turn, which the per-chain gathers of bench.py do not follow:
    such runs keep the Python submission).  Equal to the non-distributed run and to --submit python, byte for byte."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = _env(WORLD_SIZE=1, RANK=0, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY=0)
    env.pop("PUTSLAM_BENCH_BACKEND", None)
    args = ["--frames", "26"] + ARGS[2:]
    outs = {}
    for name, extra in (("dist", ["--force-dist"]), ("plain", []), ("python", ["--force-dist", "--submit", "python"])):
        npy = tmp_path / (name + ".npy")
        p = subprocess.run([sys.executable, bench, "--gpus", "1", "--dump-records", str(npy)] + extra + args, env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout + p.stderr
        j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert ("ps_batch_queue_submit" in j["config"]["submit"]) == (name != "python")
        outs[name] = np.load(npy)
    assert outs["dist"].shape == (1, 25, 18) and (outs["dist"][0, :, 17] > 0).all()
    assert outs["dist"].tobytes() == outs["plain"].tobytes() == outs["python"].tobytes()


@pytest.mark.parametrize("streams", [1, 3])
def test_bench_rccl_branch_with_a_world_of_one(tmp_path, streams):
    """The RCCL branch itself (backend "nccl", device-resident payloads): `bench.py --force-dist` initialises the process
    group with ONE rank before any other GPU call and then runs the real N > 1 control flow -- parameter broadcast on the
    device, the records packed on the chains' streams, the gathers of device tensors issued on the communication stream once the host
    has seen them packed, the fence.  The records rank 0 'gathers' must equal the non-distributed run's byte for byte.  (the only
    sequential step, pose composition PUTSLAM.cpp:735-740, happens on rank 0 after this gather.)"""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = _env(WORLD_SIZE=1, RANK=0, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(),
               HSA_ENABLE_IPC_MODE_LEGACY=0)
    env.pop("PUTSLAM_BENCH_BACKEND", None)
    dist_npy, plain_npy = tmp_path / "dist.npy", tmp_path / "plain.npy"
    p = subprocess.run([sys.executable, bench, "--gpus", "1", "--force-dist", "--streams", str(streams), "--dump-records",
                        str(dist_npy)] + ARGS, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    # rank 0's stdout is ONE line, the JSON line: RCCL prints a version block to the C-level stdout of every process that creates a
    # process group, which bench.py keeps away from it (file descriptor 1 points at stderr for the run)
    assert len(p.stdout.splitlines()) == 1 and p.stdout.startswith("{"), p.stdout[-600:]
    j = json.loads(p.stdout)
    assert j["config"]["backend"] == "nccl" and j["config"]["world_size"] == 1 and j["config"]["force_dist"] is True
    assert j["n_gpus"] == 1 and j["value"] > 0
    q = subprocess.run([sys.executable, bench, "--gpus", "1", "--streams", str(streams), "--dump-records", str(plain_npy)] + ARGS,
                       env=env, capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stdout + q.stderr
    got, want = np.load(dist_npy), np.load(plain_npy)
    assert got.shape == want.shape == (1, 13, 18)
    assert got.tobytes() == want.tobytes()
    assert np.abs(got[0, :, :16]).sum() > 0 and (got[0, :, 17] > 0).all()      # real records: poses and match counts


def test_bench_launches_its_own_ranks_when_started_plainly(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset -- the way the driver starts the one-GPU bench -- must not exit with
    rc 2 (round 4) but start the two ranks itself (fresh child processes, the launcher never touches the GPU) and pass rank 0's
    JSON line through.  Here both ranks share the one GPU over gloo; on a node they are one per GPU over RCCL."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = _env(PUTSLAM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=0)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    multi = tmp_path / "self.npy"
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--dump-records", str(multi)] + ARGS, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["pairs_per_step"] == 26 and j["value"] > 0
    got = np.load(multi)
    assert got.shape == (2, 13, 18) and got[0].tobytes() != got[1].tobytes()


def test_pack_records_device_equals_the_torch_packing(ctx):
    """ps_pack_records_device (include/putslam_hip.h): the 72-byte per-pair records of the gather in one launch -- the bytes of
    sharding.pack_records, rows beyond `valid` zero-filled."""
    import torch
    from putslam_amd import sharding, synth
    from putslam_amd._abi import EST_FIXED, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs
    seq = synth.make_sequence(12, 500, config=3, index=31337)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    cfg, _ = make_config(EST_FIXED, 512, seed=9)
    run_pairs(ctx, default_ransac_params(REPROJECTION_ERROR), cfg, TUM_FR1_K, fs, pb)
    torch.cuda.synchronize()
    P = len(seq["pairs"])
    st32 = pb.stats.view(torch.int32).view(P, -1)
    want = sharding.pack_records(pb.pose, st32[:, 5], st32[:, 0])
    got = sharding.pack_records_device(ctx, pb.pose, st32, pad_to=P + 3)
    torch.cuda.synchronize()
    assert got.shape == (P + 3, 18) and got[:P].cpu().numpy().tobytes() == want.cpu().numpy().tobytes()
    assert float(got[P:].abs().sum()) == 0.0 and float(got[:P, :16].abs().sum()) > 0
    part = sharding.pack_records_device(ctx, pb.pose[2:7], st32[2:7])
    torch.cuda.synchronize()
    assert part.cpu().numpy().tobytes() == want[2:7].cpu().numpy().tobytes()
