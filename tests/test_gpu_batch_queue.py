"""PsBatchQueue (include/putslam_hip.h): the multi-chain submission inside the library (four chains by default).  Whatever chain a pair runs on, the outputs
are those of ONE ps_vo_pairs_device call -- checked against the oracle's Matcher::match data flow (reference
src/Matcher/matcher.cpp:470-515) and against the single call, byte for byte."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from putslam_amd import api, synth
from putslam_amd._abi import (EST_FIXED, EST_RANSAC, EST_USAC, EUCLIDEAN_ERROR, REPROJECTION_ERROR, TUM_FR1_K,
                              default_ransac_params, make_config)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from test_gpu_batch import _compare  # noqa: E402


@pytest.mark.parametrize("chains", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("split_from", [None, 20])
@pytest.mark.parametrize("mode,est,H,frames", [(REPROJECTION_ERROR, EST_FIXED, 1024, 41), (EUCLIDEAN_ERROR, EST_RANSAC, 487, 30),
                                               (EUCLIDEAN_ERROR, EST_USAC, 800, 9)])
def test_queue_equals_oracle(ctx, oracle, monkeypatch, chains, split_from, mode, est, H, frames):
    """Batches go to the chains in turn, whole (the default), or -- PUTSLAM_HIP_QUEUE_SPLIT_FROM, rounds 3 - 5's recipe kept for A/B
    runs -- are split 45 % / 55 % from that many pairs on: the oracle's bytes either way."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_queue
    if split_from is None:
        monkeypatch.delenv("PUTSLAM_HIP_QUEUE_SPLIT_FROM", raising=False)
    else:
        monkeypatch.setenv("PUTSLAM_HIP_QUEUE_SPLIT_FROM", str(split_from))
    seq = synth.make_sequence(frames, 500, config=3, index=500 + mode * 10 + est)
    P = len(seq["pairs"])
    prm = default_ransac_params(mode)
    cfg, _ = make_config(est, H, seed=777)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pbs = [PairBatchDevice(seq["pairs"], fs.max_kpts) for _ in range(max(3, chains + 1))]     # consecutive batches run side by side: a block each
    q = api.BatchQueue(ctx, chains)
    assert q.chains == chains
    c = oracle.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], seq["pairs"], threads=4)
    tickets = []
    for n, pb in enumerate(pbs):
        tickets.append(run_pairs_queue(q, prm, cfg, TUM_FR1_K, fs, pb))
        b = q.last_split()
        assert b[0] == 0 and b[-1] == P and all(b[i] <= b[i + 1] for i in range(chains))
        if chains > 1 and split_from is not None and P >= split_from:
            assert b == ([0, P * 450 // 1000, P] if chains == 2 else [P * i // chains for i in range(chains + 1)])    # the split form
        else:
            assert b[n % chains + 1] - b[n % chains] == P                                                               # whole, on chain n mod chains
    for t, pb in zip(tickets, pbs):
        q.wait(t)
        assert q.query(t)
        _compare(pb.download(), c, P)
    q.close()


def test_queue_many_batches_tickets_and_stream_wait(ctx, oracle):
    """Consecutive batches into alternating output blocks, never joined; tickets older than the ring; a host stream waits for a
    batch on the device side and copies its poses: they are the single call's."""
    import torch
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs, run_pairs_queue
    seq = synth.make_sequence(26, 400, config=3, index=901)
    small = synth.make_sequence(4, 400, config=3, index=902)
    prm = default_ransac_params(REPROJECTION_ERROR)
    fs, fsS = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"]), FrameSetDevice(small["desc"], small["pts"], small["nkpts"])
    outs = [PairBatchDevice(seq["pairs"], fs.max_kpts) for _ in range(2)]
    outS = PairBatchDevice(small["pairs"], fsS.max_kpts)
    q = api.BatchQueue(ctx, 2)
    tickets = []
    for i in range(70):                                   # more than the 64 tickets of the ring: submit's flow control
        cfg, _ = make_config(EST_FIXED, 768, seed=1000 + (i % 2))
        tickets.append(run_pairs_queue(q, prm, cfg, TUM_FR1_K, fs, outs[i % 2]))
        if i % 7 == 3:                                    # small batches between the large ones go to the chains in turn
            cfgS, _ = make_config(EST_RANSAC, 487, seed=5)
            tickets.append(run_pairs_queue(q, prm, cfgS, TUM_FR1_K, fsS, outS))
    assert tickets == list(range(len(tickets)))
    side = torch.cuda.Stream()
    q.wait_on_stream(tickets[-1], side.cuda_stream)
    with torch.cuda.stream(side):
        pose_copy = outs[1].pose.clone()                  # (the last large batch, number 69, wrote outs[1])
    side.synchronize()
    q.wait(tickets[0])                                    # older than the ring: complete, returns at once
    assert q.query(tickets[0])
    with pytest.raises(api.PsError):
        q.wait(len(tickets))                              # no such ticket
    q.synchronize()
    for k in range(2):
        cfg, _ = make_config(EST_FIXED, 768, seed=1000 + k)
        ref = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, ref)
        r, g = ref.download(), outs[k].download()
        _compare(g, r, len(seq["pairs"]))
        if k == 1:
            assert pose_copy.cpu().numpy().tobytes() == r["pose"].tobytes()
    cfgS, _ = make_config(EST_RANSAC, 487, seed=5)
    c = oracle.vo_pairs(prm, cfgS, TUM_FR1_K, small["desc"], small["pts"], small["nkpts"], small["pairs"], threads=2)
    _compare(outS.download(), c, len(small["pairs"]))
    q.close()


def test_batches_sharing_an_output_block_are_kept_in_order(ctx):
    """Whole batches run side by side on different chains; a host that reuses ONE output block for consecutive batches (different
    seeds: different results) must still find the LAST batch's results there -- the queue queues such a batch behind the one that
    is in flight on the other chain instead of letting them race."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs, run_pairs_queue
    seq = synth.make_sequence(40, 600, config=3, index=4711)
    prm = default_ransac_params(REPROJECTION_ERROR)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    out = PairBatchDevice(seq["pairs"], fs.max_kpts)
    q = api.BatchQueue(ctx, 2)
    for rnd in range(5):
        chains_used = set()
        for b in range(6):
            cfg, _ = make_config(EST_FIXED, 1500, seed=50 + 10 * rnd + b)
            t = run_pairs_queue(q, prm, cfg, TUM_FR1_K, fs, out)
            sp = q.last_split()
            chains_used.add([i for i in range(2) if sp[i + 1] > sp[i]][0])
        q.wait(t)
        assert q.query(t)
        ref = PairBatchDevice(seq["pairs"], fs.max_kpts)
        run_pairs(ctx, prm, cfg, TUM_FR1_K, fs, ref)
        _compare(out.download(), ref.download(), len(seq["pairs"]))
        assert len(chains_used) == 1                      # back to back on one block: one chain
        q.synchronize()
    q.close()


def test_queue_inherits_options_and_reports_errors(ctx):
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_queue
    c2 = api.Context(0)
    c2.set_option("matcher", 0)
    c2.set_option("prune", 0)
    q = api.BatchQueue(c2, 2)
    assert all(c.get_option("matcher") == 0 and c.get_option("prune") == 0 for c in q.contexts)
    assert c2.get_option("hw_queues_seen") >= 1           # the library's default (16) or what the host set
    seq = synth.make_sequence(25, 300, config=3, index=3)
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"])
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts)
    prm = default_ransac_params(REPROJECTION_ERROR)
    bad, _ = make_config(EST_FIXED, 0, seed=1)            # numHypotheses out of range: the chain's error comes back
    with pytest.raises(api.PsError) as e:
        run_pairs_queue(q, prm, bad, TUM_FR1_K, fs, pb)
    assert "chain 0" in str(e.value) and "numHypotheses" in str(e.value)
    with pytest.raises(api.PsError):
        api.BatchQueue(c2, 9)
    good, _ = make_config(EST_FIXED, 512, seed=1)         # the queue works on after a failed batch
    t = run_pairs_queue(q, prm, good, TUM_FR1_K, fs, pb)
    q.wait(t)
    assert int(pb.download()["stats"]["accepted"].sum()) > 0
    q.close()
    c2.close()


def test_cpp_host_loop_without_the_environment_variable():
    """demos/cpp/demo_batch_queue: a C++ host that links the library and loops over submits -- GPU_MAX_HW_QUEUES unset in its
    environment: the library's constructor sets it before the process' first HIP call -- gets byte-identical results."""
    exe = os.path.join(ROOT, "demos", "cpp", "demo_batch_queue")
    assert os.path.exists(exe), "demos/cpp/demo_batch_queue is not built (__graft_entry__.build())"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    p = subprocess.run([exe, "--frames", "60", "--kpts", "800", "--hyp", "1024", "--steps", "3", "--warmup", "1", "--repeats", "2", "--check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "check against one ps_vo_pairs_device call: equal" in p.stdout
    assert "hw_queues_seen 16 (GPU_MAX_HW_QUEUES=16)" in p.stdout
    env["GPU_MAX_HW_QUEUES"] = "2"                        # a host's own value is kept, and the queue says what it means
    p = subprocess.run([exe, "--frames", "30", "--kpts", "500", "--hyp", "512", "--steps", "2", "--warmup", "1", "--repeats", "1", "--check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "hw_queues_seen 2 (GPU_MAX_HW_QUEUES=2)" in p.stdout


def fuzz_queue(iters, seed, verbose=False):
    """Random frame sets (ragged, empty frames), pair lists, schedules, metrics, chain counts and both submission forms (whole
    batches in turn / split) through a PsBatchQueue, several batches in flight with output blocks in turn, against ONE
    ps_vo_pairs_device call per batch on a context of its own.  Returns the number of configurations with a difference."""
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs, run_pairs_queue
    from test_gpu_batch import STAT_FIELDS
    rng = np.random.default_rng(seed)
    parent, ref_ctx = api.Context(0), api.Context(0)
    bad = 0
    for it in range(iters):
        F = int(rng.integers(2, 70))
        cap = int(rng.choice([64, 200, 333, 700, 1200]))
        seq = synth.make_sequence(F, cap, config=3, index=int(rng.integers(0, 2 ** 31)), inlier_frac=float(rng.uniform(0.1, 0.9)),
                                  noise=float(10 ** rng.uniform(-3.5, -1.8)))
        nk = seq["nkpts"].copy()
        if rng.random() < 0.5:
            for f in rng.integers(0, F, max(1, F // 5)):
                nk[f] = int(rng.integers(0, cap + 1))
        P = int(rng.integers(1, 3 * F))
        pairs = rng.integers(0, F, (P, 2)).astype(np.int32)
        mode = int(rng.choice([0, 1, 2, 4]))
        est, H = [(EST_RANSAC, 487), (EST_USAC, int(rng.integers(300, 3000))), (EST_FIXED, int(rng.integers(257, 5000)))][int(rng.integers(0, 3))]
        prm = default_ransac_params(mode)
        chains = int(rng.choice([1, 2, 3, 4, 4, 4, 6, 8]))
        sf = rng.choice([None, 2, 20, 60])
        if sf is None:
            os.environ.pop("PUTSLAM_HIP_QUEUE_SPLIT_FROM", None)
        else:
            os.environ["PUTSLAM_HIP_QUEUE_SPLIT_FROM"] = str(int(sf))
        fs = FrameSetDevice(seq["desc"], seq["pts"], nk)
        q = api.BatchQueue(parent, chains)
        nb = int(rng.integers(1, 7))
        outs = [PairBatchDevice(pairs, cap) for _ in range(nb)]
        seeds = [int(rng.integers(0, 2 ** 40)) for _ in range(nb)]
        tickets = []
        for b in range(nb):
            cfg, _ = make_config(est, H, seed=seeds[b])
            tickets.append(run_pairs_queue(q, prm, cfg, TUM_FR1_K, fs, outs[b]))
        ok = True
        for b in rng.permutation(nb):
            q.wait(tickets[b])
            cfg, _ = make_config(est, H, seed=seeds[b])
            ref = PairBatchDevice(pairs, cap)
            run_pairs(ref_ctx, prm, cfg, TUM_FR1_K, fs, ref)
            g, r = outs[b].download(), ref.download()
            same = np.array_equal(g["numMatches"], r["numMatches"]) and g["pose"].tobytes() == r["pose"].tobytes()
            for p in range(P):
                n = int(r["numMatches"][p])
                same = same and g["matches"][p, :n].tobytes() == r["matches"][p, :n].tobytes() and np.array_equal(g["inlierMask"][p, :n], r["inlierMask"][p, :n])
                for f in STAT_FIELDS:
                    a_, b_ = g["stats"][p][f], r["stats"][p][f]
                    same = same and (a_ == b_ or (np.isnan(a_) and np.isnan(b_)))
            if not same:
                ok = False
                if verbose:
                    print("MISMATCH", it, dict(F=F, cap=cap, P=P, mode=mode, est=est, H=H, chains=chains, split_from=sf, batch=int(b), batches=nb))
        q.close()
        bad += 0 if ok else 1
    os.environ.pop("PUTSLAM_HIP_QUEUE_SPLIT_FROM", None)
    parent.close()
    ref_ctx.close()
    return bad


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_queue_slice(seed):
    assert fuzz_queue(12, 9100 + seed, verbose=True) == 0


if __name__ == "__main__":
    import sys
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    b = fuzz_queue(n, seed, verbose=True)
    print(f"queue fuzz done: {n} configurations, {b} mismatches")
    sys.exit(1 if b else 0)
