"""Host-side logic on CPU: synthetic generator, sharding + gather (gloo, world_size 2), VO composition."""
import os
import socket
import sys

import numpy as np
import pytest

from putslam_amd import sharding, synth
from putslam_amd._abi import EST_RANSAC, EUCLIDEAN_ERROR, TUM_FR1_K, default_ransac_params, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_is_deterministic_and_shaped():
    a1, b1 = synth.make_pair(300, config=2, index=5)
    a2, b2 = synth.make_pair(300, config=2, index=5)
    assert a1["desc"].tobytes() == a2["desc"].tobytes() and b1["pts"].tobytes() == b2["pts"].tobytes()
    a3, _ = synth.make_pair(300, config=2, index=6)
    assert a1["desc"].tobytes() != a3["desc"].tobytes()
    assert a1["desc"].shape == (300, 32) and a1["pts"].shape == (300, 3) and a1["pts"].dtype == np.float32
    miss = (a1["pts"][:, 2] == 0).mean()
    assert 0.03 < miss < 0.2                                  # ~10 % missing depth exercises the z < 0.1 filter
    T = b1["T_prev_from_cur"]
    assert np.linalg.norm(T[:3, 3]) <= 0.05 + 1e-9            # below the 0.1 m gate of PUTSLAM.cpp:735-737
    link = b1["truth"] >= 0
    assert 0.6 < link.mean() < 0.8
    d = np.unpackbits(b1["desc"][link] ^ a1["desc"][b1["truth"][link]], axis=1).sum(1)
    assert 10 < d.mean() < 30                                 # ~8 % flipped bits
    seq = synth.make_sequence(5, 200, config=3, index=1)
    assert seq["desc"].shape == (5, 200, 32) and seq["pairs"].tolist() == [[0, 1], [1, 2], [2, 3], [3, 4]]


def test_shard_ranges():
    for total in (0, 1, 7, 499, 3992):
        for world in (1, 2, 3, 8):
            chunks = [sharding.shard_range(total, world, r) for r in range(world)]
            assert chunks[0][0] == 0 and chunks[-1][1] == total
            assert all(chunks[i][1] == chunks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in chunks]
            assert max(sizes) - min(sizes) <= 1
    s = sharding.shard_sequence(500, 8, 3)
    assert s["frame_lo"] == s["pair_lo"] and s["frame_hi"] == s["pair_hi"] + 1  # one-frame halo


def test_compose_trajectory_gate_and_format():
    rng = np.random.default_rng(0)
    incs = []
    for k in range(6):
        R, t = synth.random_motion(rng)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R, t
        incs.append(T)
    incs[3][:3, 3] = [0.2, 0, 0]                               # > 0.1 m: replaced by identity
    traj = sharding.compose_trajectory(np.stack(incs))
    assert np.array_equal(traj[0], np.eye(4))
    assert np.array_equal(traj[4], traj[3])                    # gated increment
    ref = np.eye(4)
    for k, T in enumerate(incs):
        if k != 3:
            ref = ref @ T.astype(np.float64)
    assert np.abs(traj[-1] - ref).max() < 1e-5
    x, y, z, w = sharding.rotation_to_quaternion_f32(np.eye(3))
    assert (x, y, z, w) == (0, 0, 0, 1)
    Rz = np.float32([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    q = sharding.rotation_to_quaternion_f32(Rz)
    assert np.allclose(q, [0, 0, np.sqrt(0.5), np.sqrt(0.5)], atol=1e-6)
    Rx = np.diag(np.float32([1, -1, -1]))                      # trace = -1: the "else" branch
    assert np.allclose(np.abs(sharding.rotation_to_quaternion_f32(Rx)), [1, 0, 0, 0], atol=1e-6)
    line = sharding.format_tum_line(1305031102.175304, traj[2])
    parts = line.split()
    assert len(parts) == 8 and parts[0] == "1305031102.1753039"  # setprecision(17), PUTSLAM.cpp:1009


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import oracle_py as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank owns an independent sequence (BASELINE config 4 in miniature); the oracle stands in for the
    # per-rank GPU work because this test runs without a GPU -- what is under test is the sharding + gather.
    seq = synth.make_sequence(4, 128, config=4, index=rank)
    prm = default_ransac_params(EUCLIDEAN_ERROR)
    cfg, _ = make_config(EST_RANSAC, 487, seed=100 + rank)
    res = po.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"], threads=1)
    # rank 0's parameter block is authoritative: the other rank starts from different values and must end up equal
    mine = default_ransac_params(EUCLIDEAN_ERROR)
    if rank != 0:
        mine.inlierThresholdEuclidean = 9.0
        mine.minimalNumberOfMatches = 99
    Kmine = np.asarray(TUM_FR1_K, np.float32) * (1.0 if rank == 0 else 2.0)
    p2, K2, est2, H2, seed2 = sharding.broadcast_params(mine, Kmine, EST_RANSAC if rank == 0 else 2, 487 + rank,
                                                        0x1234_5678_9ABC_DEF0 + rank, src=0)
    assert bytes(p2) == bytes(prm) and np.array_equal(K2.ravel(), np.asarray(TUM_FR1_K, np.float32).ravel())
    assert (est2, H2, seed2) == (EST_RANSAC, 487, 0x1234_5678_9ABC_DEF0)
    rec = sharding.pack_records(res["pose"], res["stats"]["numInliers"], res["stats"]["numMatchesIn"])
    out = sharding.gather_records(rec, dst=0)
    # the asynchronous form bench.py uses (gather of step k waited for before step k+1's gather) gives the same blocks
    import torch
    bufs = [torch.zeros_like(rec) for _ in range(world)] if rank == 0 else None
    prev = None
    for it in range(3):
        if prev is not None:
            prev.wait()
        prev, got = sharding.gather_records(rec + float(it), dst=0, out=bufs, async_op=True)
    prev.wait()
    if rank == 0:
        assert all(torch.equal(g, o + 2.0) for g, o in zip(got, out))
        q.put([o.numpy().copy() for o in out])
    else:
        q.put(rec.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q0, q1 = ctx.Queue(), ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r, q in ((0, q0), (1, q1))]
    for p in procs:
        p.start()
    gathered = q0.get(timeout=120)
    own1 = q1.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(gathered) == 2 and gathered[0].shape == (3, sharding.RECORD_FLOATS)
    assert np.array_equal(gathered[1], own1)                    # rank 1's block arrived unchanged on rank 0
    assert not np.array_equal(gathered[0], gathered[1])         # different sequences
    # rank 0 composes each sequence's trajectory from the gathered increments
    for blk in gathered:
        inc = blk[:, :16].reshape(-1, 4, 4).transpose(0, 2, 1)
        traj = sharding.compose_trajectory(inc)
        assert traj.shape == (4, 4, 4) and np.all(np.isfinite(traj))


def test_bench_launcher_sets_the_rank_environment_and_returns_the_worst_exit_code(tmp_path):
    """bench.py --gpus N started plainly becomes the launcher (bench.spawn_ranks): N child processes with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, rank 0's stdout passed through, the worst exit code returned.  The children here are a stand-in
    script (no GPU in this container): what is under test is the launcher, which itself never imports torch."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "bench.py"
    src = open(os.path.join(root, "bench.py")).read()
    # the launcher half of bench.py verbatim, its main() replaced by a child that reports its environment
    head = src[:src.index("def main():")]
    fake.write_text(head + textwrap.dedent('''
        def main():
            args = parse()
            if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
                assert "torch" not in sys.modules
                sys.exit(spawn_ranks(args))
            r = int(os.environ["RANK"])
            print("rank", r, os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])
            sys.exit(3 if (r == 1 and args.steps == 99) else 0)
        if __name__ == "__main__":
            main()
    '''))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    ok = subprocess.run([sys.executable, str(fake), "--gpus", "4", "--steps", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr
    lines = ok.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].split()[:5] == ["rank", "0", "0", "4", "127.0.0.1"]      # only rank 0's stdout comes through
    bad = subprocess.run([sys.executable, str(fake), "--gpus", "2", "--steps", "99"], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 3
