#!/usr/bin/env python3
"""bench.py -- frame-pairs/s through match + RANSAC + Kabsch/Umeyama on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (ps_vo_pairs_device: Hamming match -> cross-check ->
depth filter -> H Umeyama hypotheses scored on all matches -> selection -> refit -> final
inliers) over one batch of synthetic frame pairs that is already resident in HBM.

Workload (BASELINE.json configs[2], the 2000-keypoint throughput case of the metric; configs[3]
= the same per GPU at N > 1): one TUM-fr1-style synthetic sequence of 500 frames = 499 frame
pairs per rank, 2000 keypoints per frame, 256-bit descriptors, H = 4096 hypotheses per pair
(fixed, adaptive stop disabled), reprojection error (errorVersion 1, north_star), shipped
thresholds.  Every rank owns its own sequence (weak scaling); per-pair records (pose + counts,
72 B) are gathered to rank 0 with RCCL when N > 1: packed on the chain behind the step's kernels, the gather issued on a
communication stream once the host has seen them packed (all of a region's gathers are complete at its closing fence).

Submission: every step is ONE ps_batch_queue_submit (include/putslam_hip.h): the library owns --streams launch chains (default 4:
contexts + HIP streams; 2 read 601 k, 3 and 4 609 - 610 k, profiles/r06u) and hands the steps' batches to them in turn, whole -- consecutive steps run side by side (one chain's
matrix-core Hamming sweep beside the other's vector scoring stages) and write output blocks of their own; nothing joins the
chains, every step is complete at the closing barrier + synchronize that brackets the timed region.  (Rounds 3 - 5 split every
step's pairs 45 % / 55 % over two contexts from here: --submit python, or PUTSLAM_HIP_QUEUE_SPLIT_FROM=20 for the same inside the
library; whole batches in turn read 608 k against 559 k, profiles/r06h/queue_split_vs_turns.txt.)

The timed region of K steps is run --repeats times (default 5, each bracketed by barrier + synchronize, max over
ranks); `value` / `ms_per_step` are the MEDIAN region, `value_min` / `value_max` the spread, `timed_regions_ms_per_step`
all of them.

After the timed regions (never part of `value`) rank 0 of a single-GPU run adds short legs:
  * a single-chain leg -- the same step as ONE launch chain on one stream, HIP events around every kernel: the
    kernels' own durations.  `kernel_ms`, `roofline`, `kernel_bounds`, `single_chain` come from it (the timed region's
    per-launch figures, measured while the chains share the CUs, are kept as `timed_region_kernel_ms`);
  * one pass with the fast scoring kernel's statistics on (`score_parked_frac`; `score_evals_frac` = share of the
    complete hypotheses x matches sweep the staged scoring still evaluates, 1.0 without it);
  * `other_modes`: the same sequence as ONE launch chain in the regimes every shipped reference config runs --
    errorVersion 0 with H = 4096 fixed, and errorVersion 0 with the reference's own adaptive <= 487-iteration schedule
    (RANSAC.cpp:30,450-453) -- `ms_per_step`, `pairs_per_s` and the kernels' own durations; the timed workload with the
    staged scoring off; USAC at the reference's cap of 850 000 hypotheses; `streamed*` (BASELINE configs[2] as written: frames
    uploaded from pinned host memory, results downloaded, ps_vo_stream_push_many / pop_many); `stress/E0`, `stress/E1`
    (configs[4]); `.../inliers40`, `.../inliers90` (the timed step on data with other shares of true correspondences);
    `latency` (configs[1] through the C ABI, demos/cpp/demo_latency);
  * `cpu_baseline`: the oracle on the same workload on all host cores, 5 passes, median, with the SIMD popcount
    matcher (OpenCV's normHamming is vectorised; the scalar-popcnt figure is kept as `scalar_matcher_value`)
    (+ the reference's own <= 487-iteration schedule as `cpu_reference_schedule`).

Rank 0 prints ONE JSON line (see the field notes in DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TOPS = 78.6           # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (one wave64 instruction per SIMD every 2 cycles)
MFMA_FP4_PEAK_TFLOPS = 10066.0  # dense FP4: 32x32x64 per 32 cycles per SIMD x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md)
# VALU instructions per unit of the hot loops (unit = one descriptor pair / one (hypothesis, match) evaluation per lane), counted
# in the ISA: read from profiles/isa_mix.json (regenerate with profiles/isa_mix.py); the constants below are only the fallback
# when that file is missing.  PK_PER_UNIT: of those, packed two-lane f32 instructions (v_pk_*_f32: two f32 operations per
# lane, issued over 4 cycles -- the same f32 rate as two plain instructions at 2 cycles): counted twice in `lane_ops`
VALU_PER_UNIT = {"ps_hamming_nn": 17.5, "ps_ransac_score_exact<0>": 15, "ps_ransac_score_exact<1>": 61,
                 "ps_ransac_score_exact<4>": 15, "ps_ransac_score_exact<2>": 71,
                 "ps_ransac_score_fast<0>": 9.5, "ps_ransac_score_fast<4>": 11.5,
                 "ps_ransac_score_fast<1>": 23, "ps_ransac_score_fast<2>": 40}
PK_PER_UNIT = {"ps_ransac_score_fast<1>": 16, "ps_ransac_score_fast<2>": 16, "ps_ransac_score_fast<0>": 9.0,
               "ps_ransac_score_fast<4>": 9.0}
ISA_MIX_SOURCE = "built-in fallback constants"


def _load_isa_mix():
    global ISA_MIX_SOURCE
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
        for m in (1, 2):
            e = j["ps_ransac_score_fast<%d> stage1" % m]
            VALU_PER_UNIT["ps_ransac_score_fast<%d>" % m] = e["valu"]
            PK_PER_UNIT["ps_ransac_score_fast<%d>" % m] = e["packed"]
        for m in (0, 4):
            e = j["ps_ransac_score_euclid<%d>" % m]
            VALU_PER_UNIT["ps_ransac_score_fast<%d>" % m] = e["valu_per_match"]
            PK_PER_UNIT["ps_ransac_score_fast<%d>" % m] = e["packed_per_match"]
        for m in (0, 1, 2, 4):
            VALU_PER_UNIT["ps_ransac_score_exact<%d>" % m] = j["ps_ransac_score<%d>" % m]["valu_per_match"]
        VALU_PER_UNIT["ps_hamming_nn"] = j["ps_hamming_nn"]["valu_per_pair"]
        ISA_MIX_SOURCE = "profiles/isa_mix.json"
    except (OSError, KeyError, ValueError):
        pass


_load_isa_mix()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=500, help="frames per sequence (pairs = frames - 1)")
    ap.add_argument("--kpts", type=int, default=2000)
    ap.add_argument("--hyp", type=int, default=4096)
    ap.add_argument("--error-version", type=int, default=1, help="RANSAC::ERROR_VERSION (0 Euclid, 1 reprojection)")
    ap.add_argument("--estimator", default="fixed", choices=["fixed", "ransac", "usac"])
    ap.add_argument("--preset", default=None, choices=["demoMatching", "sequence", "stress"],
                    help="BASELINE configs: demoMatching = configs[1] (one pair per step), sequence = configs[2] "
                         "(default), stress = configs[4] (5000 kpts, H = 100000, 8 pairs per step)")
    ap.add_argument("--streams", type=int, default=4,
                    help="sub-batch chains of the step, one HIP stream and one context (scratch arena) each; with "
                         "--join end they run freely, so one chain's popcount sweep overlaps another's scoring sweep")
    ap.add_argument("--join", default="end", choices=["step", "end"],
                    help="step: every step forks from and joins the default stream; end: the sub-batch chains are ordered "
                         "only within their own stream, consecutive steps pipeline, one synchronisation at the fence")
    ap.add_argument("--split", type=float, default=0.45,
                    help="with --streams 2: fraction of the pairs on stream 0.  Unequal halves keep the two chains out of step -- one "
                         "runs its matrix-core Hamming sweep while the other is in its vector scoring sweep -- where equal ones march in "
                         "lockstep: 0.45 reads 557 - 561 k, 0.5 541 - 549 k (profiles/r05k/chains_ab.txt)")
    ap.add_argument("--submit", default="queue", choices=["queue", "python"],
                    help="queue (default): every step is ONE ps_batch_queue_submit -- the library owns the chains (contexts + streams), "
                         "splits the batch 45 %% / 55 %% and never joins them; python: rounds 3 - 5's submission, the sub-batches handed "
                         "to --streams contexts from here (also taken for --cuts, --join step, another --split, --shard sequence)")
    ap.add_argument("--no-native-legs", action="store_true", help="skip the C++ host legs (demos/cpp/demo_batch_queue, demo_sequences_multi_gpu)")
    ap.add_argument("--cuts", default=None, help="explicit cut points of the chains as fractions of the pairs, comma separated (streams - 1 values)")
    ap.add_argument("--dump-records", default=None,
                    help="test hook: write the per-pair records rank 0 holds after the last step (numpy .npy)")
    ap.add_argument("--as-rank", type=int, default=None,
                    help="test hook: single-rank run with the sequence and seed of this rank of a multi-rank run")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps each; value = the median region, value_min/max = the spread")
    ap.add_argument("--shard", default="pairs", choices=["pairs", "sequence"],
                    help="N > 1: 'pairs' = one sequence per GPU (BASELINE configs[3], weak scaling); 'sequence' = ONE "
                         "sequence of --frames frames split over the ranks with a one-frame halo "
                         "(sharding.shard_sequence; strong scaling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 control flow (RCCL process group, parameter broadcast, per-step asynchronous gather "
                         "of device-resident records, fence) even with ONE rank: the only way to execute that branch on a "
                         "one-GPU box (also PUTSLAM_BENCH_FORCE_DIST=1)")
    ap.add_argument("--warm-seconds", type=float, default=1.0,
                    help="after the --warmup steps, keep stepping (untimed) until this much wall time has passed: the timed "
                         "regions start on a chip at its steady clock")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the errorVersion-0 legs after the timed regions")
    ap.add_argument("--no-streamed", action="store_true", help="skip the streamed legs (frames from pinned host memory)")
    ap.add_argument("--no-data-legs", action="store_true", help="skip the legs with 40 %% / 90 %% true correspondences")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-call latency leg (demos/cpp/demo_latency)")
    ap.add_argument("--no-stress", action="store_true", help="skip the configs[4] legs (5000 keypoints, H = 100 000)")
    ap.add_argument("--stream-chunk", type=int, default=250, help="frames per chunk of the streamed leg (125 and 500 are reported beside it)")
    ap.add_argument("--stream-lanes", type=int, default=0, help="lanes of the streamed leg: launch chains that run side by side (the library's stream_ahead places queue more chunks behind them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work budget of the cpu_baseline leg")
    a = ap.parse_args()
    if a.preset == "demoMatching":
        a.frames = 2
    elif a.preset == "stress":
        a.frames, a.kpts, a.hyp = 9, 5000, 100000
    return a


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process becomes the launcher.  It never
    touches the GPU (no torch import, no HIP call): it starts N fresh child processes -- one rank per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would set them -- passes rank 0's stdout (the JSON line) through, and
    returns the worst exit code.  Nothing is exec'ed."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    failed_at = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and failed_at is None:
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 30.0:
            for p in procs:                        # a rank died: its peers wait in a collective for ever -- end exactly them
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = p.wait()
        if rc != 0:
            worst = rc if worst == 0 else worst
    return 1 if worst < 0 else worst


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # The sub-batches of a step run on separate HIP streams; with the runtime's default of four hardware queues two
    # streams can end up sharing one (then they serialise).  More queues keep them apart (measured: no effect on two
    # streams, 105 k -> 117 k pairs/s on three).  Must be set before the HIP runtime initialises.
    # (round 5: 16 -- the streamed legs run up to six lanes + an upload and a download stream beside the chains' streams; with
    # eight queues two of those shared one and the leg dropped from 390 k to 130 k pairs/s; the headline is the same either way)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    # Rank 0's stdout carries ONE line, the JSON line.  Libraries write to the C-level stdout on their own -- RCCL prints a block with
    # its version and path when the process group is created, flushed when the process ends: five more lines behind the JSON line of
    # every run with a process group --, so file descriptor 1 points at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running with the launcher's world size", file=sys.stderr)
    # PUTSLAM_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 control flow run with several ranks on ONE GPU
    # (records staged through the host); the measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("PUTSLAM_BENCH_BACKEND", "nccl")
    # --force-dist: the distributed branch with a world of one (RCCL initialised, collectives on device tensors)
    dist_on = world > 1 or args.force_dist or os.environ.get("PUTSLAM_BENCH_FORCE_DIST", "0") == "1"
    ndev = max(torch.cuda.device_count(), 1)          # (device_count does not initialise the GPU)
    dev = torch.device(f"cuda:{local_rank % ndev if backend != 'nccl' else local_rank}")
    if dist_on:
        # the process group comes FIRST, before any other GPU call of this process (RCCL picks the device up from device_id)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(dev)
    xdev = dev if backend == "nccl" else torch.device("cpu")       # where collective payloads live

    from putslam_amd import api, synth
    from putslam_amd._abi import (EST_FIXED, EST_RANSAC, EST_USAC, TUM_FR1_K, default_ransac_params, make_config)
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_queue, run_pairs_split

    est = {"fixed": EST_FIXED, "ransac": EST_RANSAC, "usac": EST_USAC}[args.estimator]
    S = max(1, args.streams)
    prm = default_ransac_params(args.error_version)
    vrank = rank if args.as_rank is None else args.as_rank   # which rank's sequence / seed this process works on
    cfg, _ = make_config(est, args.hyp, seed=0xB0B0 + vrank)

    if dist_on:
        # the run's parameter block comes from rank 0 (one ~120 B broadcast at start, outside the timed region)
        from putslam_amd import sharding as _sh
        prm, _K, est, _H, _seed = _sh.broadcast_params(prm, TUM_FR1_K, est, args.hyp, 0xB0B0, src=0, device=xdev)
        args.hyp = _H
        cfg, _ = make_config(est, args.hyp, seed=_seed + rank)
    # -- synthetic sequence of this rank (config 3; config 4 = one such sequence per GPU) --
    from putslam_amd import sharding
    shard_seq = args.shard == "sequence" and world > 1
    if shard_seq:
        # ONE sequence for the whole job; this rank uploads frames [frame_lo, frame_hi) (the last one is the halo the
        # next rank uploads too) and works on pairs [pair_lo, pair_hi); pair p keeps its global hypothesis stream seed + p
        full = synth.make_sequence(args.frames, args.kpts, config=3, index=0)
        sh = sharding.shard_sequence(args.frames, world, rank)
        f0, f1 = sh["frame_lo"], sh["frame_hi"]
        npairs = sh["pair_hi"] - sh["pair_lo"]
        seq = dict(desc=full["desc"][f0:max(f1, f0 + 1)], pts=full["pts"][f0:max(f1, f0 + 1)],
                   nkpts=full["nkpts"][f0:max(f1, f0 + 1)],
                   pairs=np.stack([np.arange(npairs), np.arange(1, npairs + 1)], axis=1).astype(np.int32).reshape(-1, 2))
        cfg, _ = make_config(est, args.hyp, seed=0xB0B0 + sh["pair_lo"])
        Pmax = -(-(args.frames - 1) // world)           # largest shard: gather blocks are padded to it
    else:
        seq = synth.make_sequence(args.frames, args.kpts, config=3, index=vrank)
    P = len(seq["pairs"])
    if not shard_seq:
        Pmax = P
    # The submission of the timed region: a PsBatchQueue of S chains -- the library owns contexts and streams and hands the steps'
    # batches to the chains in turn, whole (consecutive steps run side by side and write output blocks of their own) --, unless an
    # experiment asks for something the queue does not do.
    use_queue = (args.submit == "queue" and not args.cuts and args.join == "end" and S <= 8 and (S != 2 or abs(args.split - 0.45) < 1e-9)
                 and not shard_seq)
    queue = None
    if use_queue:
        parent = api.Context(dev.index)          # the queue's parent: options' source, error texts
        queue = api.BatchQueue(parent, S)
        ctxs = queue.contexts
    else:
        ctxs = [api.Context(dev.index) for _ in range(S)]
    ctx = ctxs[0]
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"], device=str(dev))
    # output blocks: one per chain of the queue, used in turn (step n + S runs on step n's chain, behind it)
    pbs = [PairBatchDevice(seq["pairs"], fs.max_kpts, device=str(dev)) for _ in range(S if use_queue else 1)]
    pb = pbs[0]
    bounds = [P * i // S for i in range(S + 1)]
    if S == 2:
        bounds = [0, int(P * args.split), P]
    if args.cuts:                                          # (experiments: explicit cut points, e.g. --streams 3 --cuts 0.3,0.63)
        bounds = [0] + [int(P * float(c)) for c in args.cuts.split(",")] + [P]
        assert len(bounds) == S + 1 and all(bounds[i] < bounds[i + 1] for i in range(S))
    # one non-default torch stream per sub-batch chain (the C ABI reads a NULL stream as "the context's private
    # stream", so the legacy default stream is never handed over)
    if use_queue:   # the chains' own streams, as torch sees them (record packing / gathers of a multi-rank run are queued on them)
        chains = [torch.cuda.ExternalStream(c.stream_ptr, device=dev) for c in ctxs]
        # (PUTSLAM_HIP_QUEUE_SPLIT_FROM = rounds 3 - 5's split of every batch over the chains, kept in the library for A/B runs)
        qsf = os.environ.get("PUTSLAM_HIP_QUEUE_SPLIT_FROM")
        queue_whole = not (qsf and qsf.isdigit() and P >= int(qsf) and S > 1)
        if not queue_whole:
            bounds = [0, P * 450 // 1000, P] if S == 2 else bounds
    else:
        queue_whole = False
        chains = [torch.cuda.Stream(device=dev) for _ in range(S)]
    join = args.join == "step"
    # gather blocks: chain i carries pairs [bounds[i], bounds[i+1]) of this rank; with --shard sequence the shards differ
    # by at most one pair and every rank pads its LAST chain's block to the size of the largest shard's
    gsize = [bounds[i + 1] - bounds[i] for i in range(S)]
    if queue_whole:
        gsize = [P] * S                 # a step's whole batch runs on one chain: every chain gathers blocks of P records
    if shard_seq:
        gsize = [(Pmax * (i + 1) // S) - (Pmax * i // S) for i in range(S)]
        bounds = [min(P, Pmax * i // S) for i in range(S)] + [P]
    gathered = ([[torch.zeros((gsize[i], sharding.RECORD_FLOATS), dtype=torch.float32, device=xdev)
                  for _ in range(world)] for i in range(S)] if (dist_on and rank == 0) else [None] * S)
    deferred = []            # steps whose records are being packed: (chain, record block, event behind the packing), in step order
    works = []               # gathers issued and not known complete: (work, the record block it reads)
    comm = torch.cuda.Stream(device=dev) if dist_on else None    # the gathers are issued here once their records are packed
    state = {"step": 0, "parts": [(i, bounds[i], bounds[i + 1]) for i in range(S)]}

    def step():
        # one batch per step: through the queue (whole, on the chains in turn) or as S sub-batches on S streams
        # (device_batch.run_pairs_split); either way the chains are ordered only within their own stream, so consecutive steps
        # pipeline into each other
        out = pbs[state["step"] % len(pbs)]
        state["step"] += 1
        if use_queue:
            run_pairs_queue(queue, prm, cfg, TUM_FR1_K, fs, out)
            if dist_on:
                b = queue.last_split()
                state["parts"] = [(i, b[i], b[i + 1]) for i in range(S) if b[i + 1] > b[i]]
        else:
            run_pairs_split(ctxs, chains, prm, est, args.hyp, cfg.seed, TUM_FR1_K, fs, out, bounds=bounds, join=join)
        if dist_on:
            # the only exchange of the path: 72 B per pair (pose + counts) to rank 0, RCCL gather over xGMI.  The records are packed
            # on the chain the step ran on, right behind its kernels (before that chain's next batch can overwrite the block); the
            # gather itself is issued -- on a communication stream of its own -- when the HOST has seen the packing complete, at a
            # later step or at the fence, gathers in step order on every rank.  Nothing is queued behind a wait for a running batch:
            # a cross-queue wait that stays pending costs the chains 5 % of their rate at this batch size and 19 % at 125 pairs
            # (profiles/r06v/pending_waits.txt; rounds 3 - 6 issued the gather at once, behind torch.distributed's event wait).
            st32 = out.stats.view(torch.int32).view(P, -1)           # PsRansacStats: [5] numInliers, [0] numMatchesIn
            for i, lo, hi in state["parts"]:
                # (ONE launch of the library on the chain, padded to the common block size -- --shard sequence --; torch's slice
                #  assignments were five launches between the chain's batches: 3 % of the step)
                rec = sharding.pack_records_device(ctxs[i], out.pose[lo:hi], st32[lo:hi], pad_to=gsize[i], stream=chains[i])
                with torch.cuda.stream(chains[i]):
                    ev = torch.cuda.Event()
                    ev.record(chains[i])
                deferred.append((i, rec, ev))
            issue_gathers(False)

    def issue_gathers(force):
        while deferred:
            i, rec, ev = deferred[0]
            if force:
                ev.synchronize()
            elif not ev.query():
                break
            deferred.pop(0)
            with torch.cuda.stream(comm):
                rec = rec.to(xdev)
                work, _ = sharding.gather_records(rec, dst=0, out=gathered[i], async_op=True)
            works.append((work, rec))
            while len(works) > 8 * S:                    # (an old gather: long complete; its record block may go)
                works.pop(0)[0].wait()

    def fence():
        if dist_on:
            issue_gathers(True)
            with torch.cuda.stream(comm):
                for w, _ in works:
                    w.wait()
            del works[:]
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # ... and until the chip has been busy for --warm-seconds in all (clock ramp: round 3's first timed region was 8 % slower
    # than the rest whatever --warmup said); whole steps, fenced, outside the timed regions
    warm_steps = args.warmup
    tw0 = time.perf_counter()
    warmed = 0.0
    while warmed < args.warm_seconds:
        for _ in range(20):
            step()
        fence()
        warm_steps += 20
        warmed = time.perf_counter() - tw0
        if dist_on:   # every rank must leave this loop after the same number of steps (collectives inside): agree on the clock
            tt = torch.tensor([warmed], dtype=torch.float64, device=xdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            warmed = float(tt.item())
    timed_events = os.environ.get("PUTSLAM_BENCH_TIMED_EVENTS", "0") == "1"
    if timed_events:
        # (round 4: off by default.  The HIP events around every kernel put a ~10 us bubble behind each of them on the
        # stream -- 3 % of a single chain's step; per-launch durations of the contended region are an optional extra leg now)
        for c in ctxs:
            c.enable_timing(True)                                      # HIP events on the launch stream(s)
    # --repeats timed regions of exactly --steps steps, each bracketed by barrier + synchronize on both sides; the
    # region time is the max over ranks; the line reports the median region (and the spread)
    region_s = []
    for _ in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if dist_on:
            tt = torch.tensor([el], dtype=torch.float64, device=xdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        region_s.append(el)
    elapsed = float(np.median(region_s))
    instrumented_steps = 0
    if not timed_events:
        # per-launch durations under the contended submission (and the only per-kernel figures of a multi-rank run): a few
        # extra steps with HIP events on, outside the timed regions
        for c in ctxs:
            c.enable_timing(True)
        instrumented_steps = max(2, min(5, args.steps))
        for _ in range(instrumented_steps):
            step()
        fence()
    totals = {}
    for c in ctxs:
        for kname, (ms_sum, n) in c.kernel_time_totals().items():
            a0, n0 = totals.get(kname, (0.0, 0))
            totals[kname] = (a0 + ms_sum, n0 + n)
        c.enable_timing(False)
    # ---- single-chain leg (untimed for `value`): the same step as ONE launch chain on one stream, so that every kernel
    # runs alone on the chip.  Its HIP-event durations are the kernels' own (the timed region above runs S chains
    # side by side: there a launch's duration includes the time it shares the CUs with the other chains' kernels).
    # `roofline` is computed from this leg; profiles/ holds the rocprofv3 kernel trace of `--streams 1`.
    solo = {}
    parked_frac = None
    evals_made = 0
    evals_frac = None
    if world == 1:
        c0 = ctxs[0]
        solo_steps = max(3, min(8, args.steps))

        def solo_step():
            run_pairs_split([c0], [chains[0]], prm, est, args.hyp, cfg.seed, TUM_FR1_K, fs, pb, bounds=[0, P], join=False)

        solo_step()
        torch.cuda.synchronize(dev)
        c0.enable_timing(True)
        ts0 = time.perf_counter()
        for _ in range(solo_steps):
            solo_step()
        torch.cuda.synchronize(dev)
        solo_ms_per_step = (time.perf_counter() - ts0) / solo_steps * 1e3
        solo = {k: v[0] / max(v[1], 1) for k, v in c0.kernel_time_totals().items()}
        c0.enable_timing(False)
        # one more pass with the fast scoring kernel's statistics on: the share of (hypothesis, match) evaluations
        # that fell inside the error band and were re-done by the value-exact code
        if args.error_version in (0, 1, 4) and c0.get_option("score") >= 1:
            c0.set_option("score_stats", 1)
            solo_step()
            torch.cuda.synchronize(dev)
            pk, ev = c0.score_stats()
            c0.set_option("score_stats", 0)
            parked_frac = pk / ev if ev else None
            evals_made = ev
    res = pb.download()
    if evals_made:
        # staged scoring: evaluations made / evaluations of the complete H x M sweep (whole wavefronts, like the counter)
        mv = res["stats"]["numMatchesValid"].astype(np.int64)
        complete = int(mv[mv >= max(3, prm.minimalNumberOfMatches)].sum()) * ((args.hyp + 255) // 256) * 256
        evals_frac = evals_made / complete if complete else None
    # ---- the regimes the reference's shipped configs run (errorVersion 0: resources/putslammatcherOpenCVParameters.xml:30-31
    # and all configs/*), same sequence, ONE launch chain, HIP events around every kernel; results go to a second output
    # block so that `res` above stays the timed workload's
    other_modes = None
    if world == 1 and not args.no_other_modes:
        other_modes = {}
        pb2 = PairBatchDevice(seq["pairs"], fs.max_kpts, device=str(dev))
        legs = [("E0/fixed/4096", 0, EST_FIXED, 4096, 1), ("E0/ransac/487", 0, EST_RANSAC, 487, 1)]
        if args.preset is None:
            # the timed workload itself with the staged scoring OFF (every hypothesis scored completely): what the
            # staged form saves on THIS data is a measured figure, not a derived one
            legs.append(("E%d/%s/%d/prune0" % (args.error_version, args.estimator, args.hyp), args.error_version, est, args.hyp, 0))
            # the reference's other estimator at ITS cap (USAC_wrapper.cpp:66,70: 850 000 hypotheses at most): the schedules end
            # after a handful of iterations; one call (models are parked for the leading hypotheses of every pair only)
            legs.append(("E0/usac/850000", 0, EST_USAC, 850000, 1))
        for name, ev2, est2, hyp2, prune2 in legs:
            if args.preset == "stress" and est2 == EST_FIXED:
                hyp2, name = args.hyp, "E0/fixed/%d" % args.hyp
            cfg2, _ = make_config(est2, hyp2, seed=cfg.seed)
            prm2 = default_ransac_params(ev2)
            prune_was = c0.get_option("prune")
            c0.set_option("prune", prune2 if prune_was else 0)

            def leg_step():
                run_pairs_split([c0], [chains[0]], prm2, est2, hyp2, cfg2.seed, TUM_FR1_K, fs, pb2, bounds=[0, P], join=False)

            leg_step()
            torch.cuda.synchronize(dev)
            c0.enable_timing(True)
            n_leg = 5
            tl0 = time.perf_counter()
            for _ in range(n_leg):
                leg_step()
            torch.cuda.synchronize(dev)
            leg_ms = (time.perf_counter() - tl0) / n_leg * 1e3
            kms = {k: v[0] / max(v[1], 1) for k, v in c0.kernel_time_totals().items()}
            c0.enable_timing(False)
            c0.set_option("prune", prune_was)
            st2 = pb2.download()["stats"]
            other_modes[name] = {"ms_per_step": leg_ms, "pairs_per_s": P / (leg_ms * 1e-3), "kernel_ms": kms,
                                 "kernel_ms_sum": sum(kms.values()), "steps": n_leg,
                                 "mean_iterations_run": float(st2["iterationsRun"].mean()),
                                 "mean_inliers": float(st2["numInliers"].mean()),
                                 "accepted_pairs": int(st2["accepted"].sum()), "staged_scoring": bool(prune2),
                                 "score_kernel": "fast" if c0.get_option("score") >= 1 else "exact"}
            if est2 == EST_USAC:
                other_modes[name]["arena_mib"] = c0.get_option("arena_mib")
                other_modes[name]["model_slots_per_pair"] = c0.get_option("last_model_slots")
            if name == "E0/ransac/487" and S > 1:
                # the reference's own regime submitted like the timed region: S chains on S streams, steps pipelined
                pb2s = [pb2] + [PairBatchDevice(seq["pairs"], fs.max_kpts, device=str(dev)) for _ in range(S - 1)]
                turn = [0]

                def chains_step():
                    if use_queue:     # (whole batches to the library's chains in turn, an output block each)
                        run_pairs_queue(queue, prm2, cfg2, TUM_FR1_K, fs, pb2s[turn[0] % S])
                        turn[0] += 1
                    else:
                        run_pairs_split(ctxs, chains, prm2, est2, hyp2, cfg2.seed, TUM_FR1_K, fs, pb2, bounds=bounds, join=False)
                for _ in range(3):
                    chains_step()
                torch.cuda.synchronize(dev)
                n_ch = 20
                tc0 = time.perf_counter()
                for _ in range(n_ch):
                    chains_step()
                torch.cuda.synchronize(dev)
                ch_ms = (time.perf_counter() - tc0) / n_ch * 1e3
                other_modes[name]["chains"] = {"streams": S, "ms_per_step": ch_ms, "pairs_per_s": P / (ch_ms * 1e-3), "steps": n_ch}
    # (The C++ host legs run BEFORE the streamed legs: those leave this process holding all 16 of its hardware queues, and a child
    # that wants ten more -- the shard demo: four chains, a communication stream, RCCL's own -- then shares the chip's queue slots with
    # them: 544 - 550 k instead of 590 - 597 k, profiles/r06v/native_shard_ab.txt.  A host of its own has the slots to itself.)
    if other_modes is not None and args.preset is None and not args.no_native_legs:
        # ---- what a C / C++ host that links the library gets on THIS workload: (a) a loop of ps_batch_queue_submit calls
        # (demos/cpp/demo_batch_queue), (b) the sharding layer with a world of one (demos/cpp/demo_sequences_multi_gpu: a queue per
        # member, asynchronous RCCL gather of the records) -- child processes of their own, GPU_MAX_HW_QUEUES unset in their environment
        try:
            other_modes.update(native_legs(args, seq, cfg))
        except Exception as e:
            other_modes["native"] = {"error": repr(e)}
    if other_modes is not None and args.preset is None and not args.no_streamed:
        # ---- BASELINE configs[2] as written: the 500 frames STREAMED (frames start in pinned HOST memory, every step uploads
        # all of them and downloads every pair's matches / mask / pose / stats), ps_vo_stream_push_many + pop_many
        try:
            # (a context of its own: c0 is a chain of the queue and carries the queue's side_by_side option, which the stream's
            # lanes would inherit)
            sctx = api.Context(dev.index)
            other_modes.update(streamed_legs(args, api, sctx, seq, prm, est, cfg, res, dev))
            sctx.close()
        except Exception as e:                                    # a leg must never cost the line its headline
            other_modes["streamed"] = {"error": repr(e)}
    if other_modes is not None and args.preset is None and not args.no_stress:
        # ---- BASELINE configs[4] under the driver's clock: 8 pairs x 5000 keypoints x H = 100 000, fixed schedule
        try:
            other_modes.update(stress_legs(args, api, c0, chains[0], dev, ctxs, chains, queue))
        except Exception as e:
            other_modes["stress"] = {"error": repr(e)}
    if other_modes is not None and args.preset is None and not args.no_data_legs:
        # ---- the headline's data dependence under the driver's clock: the staged scoring abandons what cannot win, so the timed
        # workload's rate depends on the share of true correspondences (70 % in the generator); the same step on sequences with
        # 40 % and 90 %, submitted like the timed region
        try:
            other_modes.update(data_legs(args, ctxs, chains, prm, est, cfg, bounds, dev, queue))
        except Exception as e:
            other_modes["data"] = {"error": repr(e)}
    if other_modes is not None and args.preset is None and not args.no_latency:
        # ---- BASELINE configs[1]: what ONE call costs a C / C++ host (demos/cpp/demo_latency: the C ABI timed with std::chrono
        # in a child process of its own -- no Python, no torch, no second stream)
        try:
            other_modes["latency"] = latency_leg()
        except Exception as e:
            other_modes["latency"] = {"error": repr(e)}
    if args.dump_records:
        # test hook (tests/test_gpu_multirank.py): the 72-byte per-pair records rank 0 holds after the last step
        if dist_on and rank == 0:
            # (the chains the LAST step ran on: all of them with the split submissions, one with the queue's whole batches)
            blocks = [np.concatenate([gathered[i][r].cpu().numpy() for i, _, _ in state["parts"]]) for r in range(world)]
            np.save(args.dump_records, np.stack(blocks))
        elif not dist_on:
            st32 = res["stats"].view(np.int32).reshape(P, -1)
            rec = sharding.pack_records(torch.from_numpy(res["pose"]), torch.from_numpy(st32[:, 5].copy()),
                                        torch.from_numpy(st32[:, 0].copy()))
            np.save(args.dump_records, rec.numpy()[None])

    if rank == 0:
        stats = res["stats"]
        P_total_per_step = (args.frames - 1) if shard_seq else P * world
        pairs_total = P_total_per_step * args.steps
        value = pairs_total / elapsed
        m_in = float(stats["numMatchesIn"].mean())
        m_valid = float(stats["numMatchesValid"].mean())
        Hs = args.hyp
        bytes_per_pair = float(np.mean([api.algorithmic_bytes(args.kpts, int(s["numMatchesIn"]),
                                                              int(s["numMatchesValid"]), Hs) for s in stats]))
        kern = {k: (v[0] / max(v[1], 1)) for k, v in totals.items()}   # timed region: average launch duration, ms
        matcher = "mfma" if ctx.get_option("matcher_used") == 1 else "valu"   # what this workload's calls ran
        matcher_fused = bool(ctx.get_option("matcher_fused")) and matcher == "mfma"
        score = ({1: "fast", 2: "mfma"}.get(ctx.get_option("score"), "exact") if args.error_version == 1 else
                 ("fast" if (args.error_version in (0, 4) and ctx.get_option("score") >= 1) else "exact"))

        def kernel_bounds(kms, pairs_per_launch):
            """What actually bounds the two sweeps (the path is compute-bound, not HBM-bound): achieved rate of the
            bounding unit / its nominal peak (MI355X_MICROARCH.md).  Instruction counts: profiles/isa_mix.json."""
            out = {}
            if "ps_hamming_mfma" in kms:
                flops = 2.0 * 256.0 * args.kpts * args.kpts * pairs_per_launch
                ach = flops / (kms["ps_hamming_mfma"] * 1e-3) / 1e12
                out["ps_hamming_mfma"] = {"bound": "mfma", "dtype": "fp4 (e2m1) x fp4 -> f32", "achieved": ach,
                                          "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": ach / MFMA_FP4_PEAK_TFLOPS}
            vk = {"ps_hamming_nn": (VALU_PER_UNIT["ps_hamming_nn"], args.kpts * float(args.kpts) * pairs_per_launch / 64.0),
                  # evaluations the scoring step really made (staged scoring: a fraction of the complete H x M sweep; the
                  # counter of the statistics pass), not the sweep it stands for
                  "ps_ransac_score": (VALU_PER_UNIT["ps_ransac_score_%s<%d>" % (score, args.error_version)]
                                      if ("ps_ransac_score_%s<%d>" % (score, args.error_version)) in VALU_PER_UNIT else None,
                                      (evals_made / 64.0 * pairs_per_launch / P) if evals_made
                                      else Hs * m_valid * pairs_per_launch / 64.0)}
            for name, (per_unit, wave_units) in vk.items():
                if name in kms and per_unit:
                    ach = wave_units * per_unit * 64.0 / (kms[name] * 1e-3) / 1e12
                    out[name] = {"bound": "valu", "valu_instructions_per_unit": per_unit, "instruction_counts_from": ISA_MIX_SOURCE, "achieved": ach,
                                 "peak": VALU_PEAK_TOPS, "unit": "T lane-instructions/s", "frac": ach / VALU_PEAK_TOPS}
                    pk = PK_PER_UNIT.get("%s_%s<%d>" % (name, score, args.error_version)) if name == "ps_ransac_score" else None
                    if pk:
                        # f32 lane OPERATIONS (a packed instruction = two): the figure to hold against the f32 vector peak
                        out[name]["lane_ops_per_unit"] = per_unit + pk
                        out[name]["achieved_lane_ops"] = ach * (per_unit + pk) / per_unit
                        out[name]["frac_lane_ops"] = out[name]["achieved_lane_ops"] / VALU_PEAK_TOPS
                    if name == "ps_ransac_score" and evals_made:
                        out[name]["note"] = ("instructions of the evaluation loops only (evaluations really made x instructions "
                                             "per evaluation; stage 1's pre-tested front costs less than that per evaluation); the "
                                             "per-hypothesis prologue (sample -> Umeyama -> SVD: measured per wavefront in "
                                             "profiles/<tag>/sq_counters_by_grid.json) and the reorder launch are inside the time "
                                             "but not in the count; `all_in` next to this has every vector instruction of the step")
                        out[name]["complete_sweep_equivalent"] = Hs * m_valid * pairs_per_launch * per_unit / (kms[name] * 1e-3) / 1e12
            return out

        rk = solo if solo else kern                      # kernels' own durations when the single-chain leg ran
        r_pairs = P if (solo or queue_whole) else P / S
        dom = max((k for k in rk if k != "ps_expand_query_fp4"), key=rk.get)
        dom_ms = rk[dom]
        achieved = bytes_per_pair * r_pairs / (dom_ms * 1e-3) / 1e9
        bounds_solo = kernel_bounds(rk, r_pairs)
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                t = json.load(open(tf))
                key = f"{args.frames}x{args.kpts}xH{args.hyp}xE{args.error_version}x{args.estimator}x{matcher}x{score}"
                traffic = t.get(key, {}).get(dom)
                # pipe-busy fractions of the same profiled workload (SQ counters of the rocprofv3 pass next to the
                # traffic pass): share of the kernel's cycles in which the vector ALU / the matrix pipe of a SIMD is busy
                sq = json.load(open(os.path.join(ROOT, os.path.dirname(t[key]["_source"]), "sq_counters.json")))
                src = os.path.dirname(t[key]["_source"]) + "/sq_counters.json (profiled run, not this one)"
                step_of = {"ps_ransac_score_fast": "ps_ransac_score", "ps_stage_reorder": "ps_ransac_score",
                           "ps_ransac_score_euclid": "ps_ransac_score", "ps_hamming_mfma_fused": "ps_hamming_mfma"}
                acc, parts = {}, {}
                for kname, c in sq.items():
                    short = kname.split("<")[0]
                    short = step_of.get(short, short)
                    if not c.get("GRBM_GUI_ACTIVE"):
                        continue
                    a_ = acc.setdefault(short, {})
                    for cn in ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
                        a_[cn] = a_.get(cn, 0.0) + float(c.get(cn, 0.0))
                    parts.setdefault(short, {})[kname] = c

                def busy(c):
                    cyc = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0      # kernel cycles x SIMDs (8 XCDs report separately)
                    return {"valu_active_frac": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / cyc,
                            "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cyc}

                for short, c in acc.items():
                    if short not in bounds_solo:
                        continue
                    # the STEP's figure: the counters of every launch of the step summed (the scoring step is six launches
                    # of four kernels; round 3's line let the last kernel of the file win)
                    pb_ = busy(c)
                    pb_["source"] = src
                    if len(parts[short]) > 1:
                        pb_["by_kernel"] = {k: busy(v) for k, v in parts[short].items()}
                    bounds_solo[short]["pipe_busy_pmc"] = pb_
                    if short in rk and c.get("SQ_INSTS_VALU"):
                        # ALL vector instructions the step issued (prologue, stages, reorder included) over the step's
                        # own duration in THIS run: the all-in rate next to the loops-only `frac`
                        allin = c["SQ_INSTS_VALU"] * 64.0 / (rk[short] * 1e-3) / 1e12
                        bounds_solo[short]["all_in"] = {"valu_wave_instructions_per_step": c["SQ_INSTS_VALU"],
                                                        "salu_instructions_per_step": c.get("SQ_INSTS_SALU"),
                                                        "achieved": allin, "unit": "T lane-instructions/s",
                                                        "frac": allin / VALU_PEAK_TOPS,
                                                        "note": "SQ_INSTS_VALU of the profiled run / this run's duration"}
            except Exception:
                pass
        # What the step needs of the chip's ISSUE capacity -- the unit that binds it: busy SIMD-cycles of the
        # profiled run (4 cycles per vector wave-instruction, SQ_ACTIVE_INST_VALU, + the matrix pipe's busy cycles, which add up
        # on gfx950) over 1024 SIMDs at 2.4 GHz, against THIS run's time per step
        issue = None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = f"{args.frames}x{args.kpts}xH{args.hyp}xE{args.error_version}x{args.estimator}x{matcher}x{score}"
            sqp = os.path.join(ROOT, os.path.dirname(t[key]["_source"]), "sq_counters.json")
            sq = json.load(open(sqp))
            busy = sum(4.0 * float(c.get("SQ_ACTIVE_INST_VALU", 0.0)) + float(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) for c in sq.values())
            floor_ms = busy / (1024.0 * 2.4e9) * 1e3
            step_ms = elapsed / args.steps * 1e3
            issue = {"bound": "issue (vector + matrix pipe, 1024 SIMDs x 2.4 GHz)", "busy_simd_cycles_per_step": busy,
                     "ms_per_step_at_full_issue": floor_ms, "ms_per_step": step_ms, "frac": floor_ms / step_ms,
                     "source": os.path.relpath(sqp, ROOT) + " (profiled single-chain run of the same workload, not this one)",
                     "note": "the two pipes' busy cycles are SUMMED: the time they need if they never overlap; with four chains they "
                             "overlap by a few per cent, so frac can read 1.00 - 1.03"}
        except Exception:
            pass
        out = {
            "metric": "frame-pairs/s (match+RANSAC+Kabsch), 640x480 @ 2000 kpts",
            "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warm_steps_before_region_1": warm_steps,
            # calls of the step in launch order (profiles/summarize.py attributes a kernel trace's launches with it)
            "launch_sequence": {"warm": warm_steps, "timed": args.steps * len(region_s), "instrumented": instrumented_steps,
                                "leg_warm": 1 if solo else 0, "leg": (solo_steps if solo else 0),
                                "stats": 1 if evals_made else 0},
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if shard_seq else "weak",
            "value_min": P_total_per_step * args.steps / max(region_s), "value_max": P_total_per_step * args.steps / min(region_s),
            "timed_regions_ms_per_step": [r / args.steps * 1e3 for r in region_s], "repeats": len(region_s),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[2]: TUM fr1/desk-style synthetic sequence, "
                             f"{args.frames} frames = {P} frame pairs per step per GPU, {args.kpts} kpts/frame, "
                             "256-bit descriptors, Hamming BF + cross-check -> depth filter -> "
                             f"H={args.hyp} 3-pt Umeyama hypotheses ({args.estimator}) scored with errorVersion "
                             f"{args.error_version} -> refit; inputs resident in HBM" +
                             ("; configs[3]: one sequence per GPU, RCCL gather of 72 B/pair to rank 0" if world > 1
                              else "")),
                "pairs_per_step": P_total_per_step, "kpts": args.kpts, "hypotheses": args.hyp,
                "world_size": (dist.get_world_size() if dist_on else 1),
                "backend": (dist.get_backend() if dist_on else None), "shard": args.shard,
                "force_dist": bool(dist_on and world == 1),
                "errorVersion": args.error_version, "estimator": args.estimator, "streams": S, "split": (args.split if S == 2 and not shard_seq else None), "join": args.join,
                "submit": (("ps_batch_queue_submit: whole batches to the library's chains in turn" if queue_whole else
                            "ps_batch_queue_submit with PUTSLAM_HIP_QUEUE_SPLIT_FROM: every batch split over the library's chains") if use_queue
                           else "python (sub-batches handed to the contexts from bench.py)"),
                "matcher_kernel": matcher + ("-fused" if matcher_fused else ""), "score_kernel": score,
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "mean_matches": m_in, "mean_valid_matches": m_valid,
                "mean_inliers": float(stats["numInliers"].mean()),
                "accepted_pairs": int(stats["accepted"].sum()),
            },
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_pair": bytes_per_pair, "pairs_per_launch": r_pairs,
                         "avg_launch_ms": dom_ms,
                         "leg": ("single launch chain (--streams 1 equivalent), run inside this process after the timed "
                                 "region: the kernel alone on the chip" if solo else "timed region"),
                         "bound_actual": bounds_solo.get(dom, {}).get("bound"),
                         "bound_actual_frac": bounds_solo.get(dom, {}).get("frac"),
                         "issue": issue,
                         "note": "the path is compute-bound (VALU issue for the scoring sweep, MFMA for the Hamming "
                                 "sweep), not HBM-bound: see kernel_bounds"},
            "kernel_ms": rk,
            "kernel_bounds": bounds_solo,
            "single_chain": ({"ms_per_step": solo_ms_per_step, "pairs_per_s": P / (solo_ms_per_step * 1e-3),
                              "kernel_ms_sum": sum(solo.values())} if solo else None),
            "timed_region_kernel_ms": kern,
            "score_parked_frac": parked_frac,
            "score_evals_frac": evals_frac,
            "staged_scoring": bool(ctx.get_option("prune")),
            "other_modes": other_modes,
            "streams_note": (None if S == 1 else
                             f"value / ms_per_step: {S} sub-batch chains on {S} HIP streams (join={args.join}); "
                             "timed_region_kernel_ms are per-launch durations of sub-batches measured while the other "
                             "chains' kernels share the CUs (they sum to more than ms_per_step); kernel_ms / roofline / "
                             "kernel_bounds come from the single-chain leg (DESIGN.md section 5)"),
        }
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU figure beside the line, on rank 0's host cores -- after the closing barrier of a multi-rank run, so that no
        # rank waits in a collective while rank 0 computes (its peers have left; their GPUs are idle, the cores are rank 0's)
        if not args.no_cpu_baseline:
            out.update(cpu_baseline(args, seq, prm, cfg, est))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


PCIE_GEN5_X16_GBS = 63.0        # PCIe 5.0 x16, one direction, after 128b/130b encoding (the MI355X host link)


def _link_rate(torch, dev, mb=64, reps=8):
    """Measured host <-> device copy rates on pinned memory, GB/s (H2D, D2H): the streamed legs' roofline."""
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    out = []
    for src, dst in ((h, d), (d, h)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        out.append(n * reps / (time.perf_counter() - t0) / 1e9)
    return out


def _stream_shape(chunk, lanes):
    """(lanes, places beyond one per lane) ps_vo_stream_configure_async uses for `lanes` = 0 and option stream_ahead = -1 (the
    defaults): by chunk size -- two lanes + one more place from 96 frames per chunk on, two + two from 48, three + three below, six
    lanes for chunks of one to four frames."""
    if lanes == 0:
        lanes = 2 if chunk >= 48 else (3 if chunk > 4 else 6)
    return lanes, (1 if chunk >= 96 else (2 if chunk >= 48 else max(0, 6 - lanes)))


def streamed_legs(args, api, c0, seq, prm, est, cfg, res, dev):
    """The timed workload with the frames STREAMED from the host (reference call shape: one frame after the other, previous
    frame kept as state, src/Matcher/matcher.cpp:452-516 in the loop of src/PUTSLAM/PUTSLAM.cpp:677-740): every step uploads
    its frames from pinned host memory in chunks, runs each chunk as one batched call on one of several lanes and downloads
    every pair's matches, mask, pose and stats.  Bound: the host link (88 KB in, 34 KB out per 2000-keypoint pair)."""
    import torch
    from putslam_amd._abi import TUM_FR1_K
    F, cap = seq["desc"].shape[:2]
    P = F - 1
    hd, hp = api.PinnedBuffer((F, cap, 32), np.uint8), api.PinnedBuffer((F, cap, 3), np.float32)
    hd.array[:] = seq["desc"]
    hp.array[:] = seq["pts"]
    nk = np.ascontiguousarray(seq["nkpts"], np.int32)
    h2d, d2h = _link_rate(torch, dev)
    out = {}

    from putslam_amd.device_batch import pack_frames
    hpk = api.PinnedBuffer((F, (cap * 44 + 15) // 16 * 16), np.uint8)     # the same frames, one block per frame (PS_FRAMES_PACKED)
    hpk.array[:] = pack_frames(seq["desc"], seq["pts"], hpk.array.shape[1])

    def run(chunk, lanes, steps, check, warm_s=0.4, results=0, windows=1, packed=True):
        st = api.VoStream(c0, cap)
        st.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=chunk, lanes=lanes, results=results, packed=packed)
        lat, sub_t = [], {}
        state = {"pairs": 0, "inl": 0, "bad": 0, "step": 0}

        def take(wait):
            b = st.pop_many(wait=wait, copy=False)
            if b is None:
                return False
            now = time.perf_counter()
            key = (b["epoch"], b["first_pair"])
            if key in sub_t:
                lat.append(now - sub_t.pop(key))
            state["pairs"] += b["count"]
            state["inl"] += int(b["stats"]["numInliers"].sum())          # the consumer reads what came back
            if check and b["epoch"] == state.get("check_epoch"):
                lo = b["first_pair"]
                if b["pose"].tobytes() != res["pose"][lo:lo + b["count"]].tobytes():
                    state["bad"] += 1
            return True

        def one_step():
            while not st.reset():
                take(True)
            ep = state["step"]
            state["step"] += 1
            f, first = 0, 0
            while f < F:
                n = min(chunk, F - f)
                t_sub = time.perf_counter()
                if (st.push_many_packed(hpk.array[f:f + n], nk[f:f + n]) if packed else
                        st.push_many(hd.array[f:f + n], hp.array[f:f + n], nk[f:f + n])):
                    pairs = n - (1 if f == 0 else 0)
                    if pairs > 0:
                        sub_t[(ep + 1, first)] = t_sub            # (epoch = resets before the block's frames)
                    first += pairs
                    f += n
                    while take(False):
                        pass
                else:
                    take(True)

        tw = time.perf_counter()
        while state["step"] < 3 or time.perf_counter() - tw < warm_s:   # (an un-warmed pipeline runs at half its rate:
            one_step()                                                   # profiles/r05a/stream_sweep.txt, first rows)
        while take(True):
            pass
        warm_steps = state["step"]
        lat.clear()
        # `windows` timed windows of `steps` steps each, the MEDIAN window is reported (one stall of a few milliseconds -- the
        # runtime growing a pool, the host descheduled -- is a fifth of a 50-ms window: round 5's single window read 340 k
        # instead of 405 k once in three runs)
        state["check_epoch"] = warm_steps + steps * windows  # the last timed step (epoch = resets before the block's frames)
        wins = []
        for _ in range(windows):
            pairs0, inl0 = state["pairs"], state["inl"]
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            while take(True):
                pass
            el_w = time.perf_counter() - t0
            wins.append((el_w, state["pairs"] - pairs0, state["inl"] - inl0))
        wins.sort(key=lambda w: w[1] / w[0])
        el, done, inl_done = wins[len(wins) // 2]
        inl0 = state["inl"] - inl_done                      # (so that the download figure below is the median window's)
        st_graphs = st.graph_launches()
        st.close()
        lat_ms = np.array(sorted(lat)) * 1e3
        leg = {"pairs_per_s": done / el, "ms_per_step": el / steps * 1e3, "steps": steps, "pairs": done, "windows": windows,
               "pairs_per_s_min": wins[0][1] / wins[0][0], "pairs_per_s_max": wins[-1][1] / wins[-1][0],
               "chunk_frames": chunk, "lanes": _stream_shape(chunk, lanes)[0], "places": sum(_stream_shape(chunk, lanes)),
               "results": ["full", "inliers", "poses"][results],
               "frames": ("packed: one block per frame, one upload per chunk (ps_vo_stream_push_many_packed)" if packed else
                          "two arrays: two uploads per chunk (ps_vo_stream_push_many)"),
               "h2d_GBps": steps * F * cap * 44 / el / 1e9,
               "d2h_GBps": (done * (cap * 17 + 108) if results == 0 else
                            (state["inl"] - inl0) * 16 + done * 108 if results == 1 else done * 108) / el / 1e9,
               "chunk_latency_ms": ({"p50": float(lat_ms[len(lat_ms) // 2]), "p95": float(lat_ms[int(len(lat_ms) * 0.95)]),
                                     "max": float(lat_ms[-1]), "n": int(len(lat_ms)),
                                     "what": "push_many call of the chunk -> its results readable on the host"} if len(lat_ms) else None)}
        if check:
            leg["equals_batched_call"] = state["bad"] == 0
        if chunk <= 4:
            leg["chunks_replayed_from_graphs"] = st_graphs
        return leg

    steps = max(10, min(40, 2 * args.steps))
    only = os.environ.get("PUTSLAM_BENCH_STREAM_LEGS")       # (diagnostic: a comma-separated subset of the side legs below)
    want = (lambda name: True) if not only else (lambda name: name in only.split(","))
    main = run(args.stream_chunk, args.stream_lanes, steps, check=1, warm_s=1.0, windows=5)   # the last step's poses against the batched call's
    main["roofline"] = {"bound": "pcie", "achieved": main["h2d_GBps"], "peak": PCIE_GEN5_X16_GBS, "unit": "GB/s",
                        "frac": main["h2d_GBps"] / PCIE_GEN5_X16_GBS, "measured_link_h2d_GBps": h2d,
                        "measured_link_d2h_GBps": d2h, "frac_of_measured": main["h2d_GBps"] / h2d,
                        "bytes_in_per_frame": cap * 44, "bytes_out_per_pair": cap * 17 + 108,
                        "note": "host -> device bytes of the frames themselves over the timed legs' wall time; uploads, kernels "
                                "and downloads of consecutive chunks overlap"}
    main["workload"] = ("BASELINE configs[2] as written: %d frames streamed from pinned host memory per step (ps_vo_stream_push_many_packed: "
                        "one block per frame, one upload per chunk; chunks of %d frames on %d lanes), every pair's matches / mask / pose / stats downloaded (pop_many); same "
                        "parameters as the timed workload" % (F, args.stream_chunk, _stream_shape(args.stream_chunk, args.stream_lanes)[0]))
    out["streamed"] = main
    # rounds 4 - 5's form, kept beside it: descriptors and points as two host arrays, two uploads per chunk
    if want("two_arrays"):
        out["streamed/two_arrays"] = run(args.stream_chunk, args.stream_lanes, steps, check=1, warm_s=0.5, windows=3, packed=False)
    # what Matcher::match itself returns -- estimatedTransformation + inlierMatches (matcher.cpp:452-516) -- instead of every
    # cross-check match + mask: a third of the download, written by a kernel straight into the pinned block
    if want("inliers"):
        out["streamed/inliers"] = run(args.stream_chunk, args.stream_lanes, steps, check=1, results=1)
    if want("poses"):
        out["streamed/poses"] = run(args.stream_chunk, args.stream_lanes, steps, check=1, results=2)   # (a host that only composes the trajectory)
    for other in (125, 250, 500):
        if other != args.stream_chunk and want("chunk%d" % other):
            out["streamed/chunk%d" % other] = run(other, args.stream_lanes, steps, check=0)
    if want("chunk32"):
        out["streamed/chunk32"] = run(32, args.stream_lanes, max(5, steps // 2), check=0, warm_s=0.6, windows=3)
    if want("chunk1"):
        out["streamed/chunk1"] = run(1, args.stream_lanes, 3, check=1, warm_s=0.2)
    if want("chunk4"):
        out["streamed/chunk4"] = run(4, args.stream_lanes, 3, check=0, warm_s=0.2)
    hd.close()
    hp.close()
    hpk.close()
    return out


def data_legs(args, ctxs, chains, prm, est, cfg, bounds, dev, queue=None):
    """The timed workload's step (same frames x keypoints, schedule, metric, the same sub-batch chains) on synthetic sequences
    with other shares of true correspondences than the generator's 70 %."""
    import torch
    from putslam_amd import synth
    from putslam_amd._abi import TUM_FR1_K
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_queue, run_pairs_split
    out = {}
    for frac in (0.4, 0.9):
        seq2 = synth.make_sequence(args.frames, args.kpts, config=3, index=0, inlier_frac=frac)
        fs2 = FrameSetDevice(seq2["desc"], seq2["pts"], seq2["nkpts"], device=str(dev))
        pb2s = [PairBatchDevice(seq2["pairs"], fs2.max_kpts, device=str(dev)) for _ in range(len(ctxs) if queue is not None else 1)]
        pb2 = pb2s[0]
        P = len(seq2["pairs"])
        turn = [0]

        def step():
            if queue is not None:     # submitted like the timed region: whole batches to the library's chains in turn
                run_pairs_queue(queue, prm, cfg, TUM_FR1_K, fs2, pb2s[turn[0] % len(pb2s)])
                turn[0] += 1
            else:
                run_pairs_split(ctxs, chains, prm, est, args.hyp, cfg.seed, TUM_FR1_K, fs2, pb2, bounds=bounds, join=False)

        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        n = max(10, args.steps)
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / n * 1e3
        st = pb2.download()["stats"]
        out["E%d/%s/%d/inliers%d" % (args.error_version, args.estimator, args.hyp, int(frac * 100))] = {
            "ms_per_step": ms, "pairs_per_s": P / (ms * 1e-3), "steps": n, "streams": len(chains),
            "true_correspondence_share": frac, "mean_inliers": float(st["numInliers"].mean()),
            "mean_matches_valid": float(st["numMatchesValid"].mean()), "accepted_pairs": int(st["accepted"].sum())}
    return out


def latency_leg(kpts=2000, calls=400):
    """One 2000-keypoint pair in the reference's own regime (errorVersion 0, <= 487 iterations) through the C ABI, as a C++
    host sees it: (a) device-resident pair, call + synchronize; (b) the same call back to back (the GPU side of the chain);
    (c) ps_vo_stream_push, host frame in / results out (Matcher::match's call shape, matcher.cpp:452-516)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "demos", "cpp", "demo_latency")
    if not os.path.exists(exe):
        return {"error": "demos/cpp/demo_latency is not built (__graft_entry__.build())"}
    p = subprocess.run([exe, str(kpts), "0", str(calls)], capture_output=True, text=True, timeout=120)
    out = {"workload": "BASELINE configs[1]: one pair, %d keypoints, errorVersion 0, RANSAC <= 487, C ABI timed from C++" % kpts,
           "rc": p.returncode}
    for key, tag in (("pair_call_and_sync_us", r"\(a\)"), ("pair_back_to_back_us", r"\(b\)"), ("pushed_frame_us", r"\(c\)")):
        m = re.search(tag + r"[^\n]*?(?:median|:) ([0-9.]+) us", p.stdout)
        out[key] = float(m.group(1)) if m else None
    m = re.search(r"\(a\)[^\n]*host time inside the call ([0-9.]+)", p.stdout)
    out["pair_host_time_in_call_us"] = float(m.group(1)) if m else None
    if out["pushed_frame_us"]:
        out["pushed_frames_per_s"] = 1e6 / out["pushed_frame_us"]
    # (d): the pipelined form at one frame per chunk, from the same C++ host
    for i, key in ((0, "pipelined_chunk1"), (1, "pipelined_chunk1_inliers")):
        m = re.search(r"\(d%d\)[^\n]*: ([0-9.]+) frames/s, result lag median ([0-9.]+) us  p90 ([0-9.]+)  \((\d+) chunks from graphs" % i, p.stdout)
        if m:
            out[key] = {"frames_per_s": float(m.group(1)), "result_lag_us_p50": float(m.group(2)), "result_lag_us_p90": float(m.group(3)),
                        "chunks_from_graphs": int(m.group(4))}
    return out


def native_legs(args, seq, cfg):
    """The timed workload through the C ABI from C++ hosts (no Python, no torch in those processes).  The sequence travels as a
    file; the children's environment has no GPU_MAX_HW_QUEUES (the library's constructor sets its default)."""
    import re
    import subprocess
    import tempfile
    out = {}
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    with tempfile.TemporaryDirectory(prefix="putslam_bench_") as td:
        path = os.path.join(td, "seq0.bin")
        F, cap = seq["desc"].shape[:2]
        with open(path, "wb") as f:
            np.array([F, cap], np.int32).tofile(f)
            np.ascontiguousarray(seq["nkpts"], np.int32).tofile(f)
            np.ascontiguousarray(seq["desc"], np.uint8).tofile(f)
            np.ascontiguousarray(seq["pts"], np.float32).tofile(f)
        common = ["--hyp", str(args.hyp), "--estimator", args.estimator, "--error-version", str(args.error_version)]
        steps = str(max(10, args.steps))
        exe = os.path.join(ROOT, "demos", "cpp", "demo_batch_queue")
        if os.path.exists(exe):
            for name, chains in (("batch_queue_cpp", max(2, args.streams)), ("batch_queue_cpp/one_context", 1)):
                p = subprocess.run([exe, "--sequence", path, "--seed", str(cfg.seed), "--steps", steps, "--repeats", "5", "--warm-seconds", "0.6", "--chains", str(chains)]
                                   + common + (["--check"] if chains >= 2 else []), capture_output=True, text=True, timeout=180, env=env)
                m = re.search(r"batch_queue: chains (\d+), median ([0-9.]+) frame-pairs/s, min ([0-9.]+), max ([0-9.]+).*hw_queues_seen (\d+) \(GPU_MAX_HW_QUEUES=([^)]*)\)", p.stdout)
                leg = {"rc": p.returncode, "what": ("demos/cpp/demo_batch_queue: a C++ loop of ps_batch_queue_submit calls over the timed workload's "
                                                     "sequence, GPU_MAX_HW_QUEUES unset in its environment" if chains >= 2 else
                                                     "the same loop through ONE context (ps_vo_pairs_device)")}
                if m:
                    leg.update({"chains": int(m.group(1)), "pairs_per_s": float(m.group(2)), "pairs_per_s_min": float(m.group(3)),
                                "pairs_per_s_max": float(m.group(4)), "hw_queues_seen": int(m.group(5)), "env_GPU_MAX_HW_QUEUES_after_load": m.group(6),
                                "steps": int(steps), "regions": 5})
                    if chains >= 2:
                        leg["equals_one_call"] = "check against one ps_vo_pairs_device call: equal" in p.stdout
                else:
                    leg["error"] = (p.stdout + p.stderr)[-400:]
                out[name] = leg
        else:
            out["batch_queue_cpp"] = {"error": "demos/cpp/demo_batch_queue is not built (__graft_entry__.build())"}
        exe = os.path.join(ROOT, "demos", "cpp", "demo_sequences_multi_gpu")
        if os.path.exists(exe):
            p = subprocess.run([exe, "--gpus", "1", "--sequence-prefix", os.path.join(td, "seq"), "--seed", str(cfg.seed), "--steps", steps, "--repeats", "5", "--warm-seconds", "0.6"]
                               + common, capture_output=True, text=True, timeout=180, env=env)
            m = re.search(r"median ([0-9.]+) frame-pairs/s in all, min ([0-9.]+), max ([0-9.]+)", p.stdout)
            leg = {"rc": p.returncode, "what": "demos/cpp/demo_sequences_multi_gpu --gpus 1: include/putslam_shard.h with a world of one -- a batch queue per "
                                               "member, records packed on the chains and gathered over RCCL asynchronously every step (ps_shard_gather_records_async), "
                                               "trajectory composed on rank 0"}
            if m:
                leg.update({"pairs_per_s": float(m.group(1)), "pairs_per_s_min": float(m.group(2)), "pairs_per_s_max": float(m.group(3)),
                            "steps": int(steps), "regions": 5})
            else:
                leg["error"] = (p.stdout + p.stderr)[-400:]
            out["native_shard"] = leg
        else:
            out["native_shard"] = {"error": "demos/cpp/demo_sequences_multi_gpu is not built (RCCL missing?)"}
    return out


def stress_legs(args, api, c0, chain, dev, ctxs=None, chains=None, queue=None):
    """BASELINE configs[4]: 8 pairs x 5000 keypoints x H = 100 000, fixed schedule (RANSAC.cpp:87-150 with
    the adaptive stop disabled), both metrics, one launch chain, HIP events around every kernel."""
    import torch
    from putslam_amd import synth
    from putslam_amd._abi import EST_FIXED, TUM_FR1_K, default_ransac_params, make_config
    from putslam_amd.device_batch import FrameSetDevice, PairBatchDevice, run_pairs_queue, run_pairs_split
    frames, kpts, hyp = 9, 5000, 100000
    seq = synth.make_sequence(frames, kpts, config=3, index=0)
    P = len(seq["pairs"])
    fs = FrameSetDevice(seq["desc"], seq["pts"], seq["nkpts"], device=str(dev))
    pb = PairBatchDevice(seq["pairs"], fs.max_kpts, device=str(dev))
    out = {}
    for ev in (0, 1):
        prm = default_ransac_params(ev)
        cfg, _ = make_config(EST_FIXED, hyp, seed=0xB0B0)

        def step():
            run_pairs_split([c0], [chain], prm, EST_FIXED, hyp, cfg.seed, TUM_FR1_K, fs, pb, bounds=[0, P], join=False)

        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / n * 1e3
        c0.enable_timing(True)
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        kms = {k: v[0] / max(v[1], 1) for k, v in c0.kernel_time_totals().items()}
        c0.enable_timing(False)
        st = pb.download()["stats"]
        bpp = float(np.mean([api.algorithmic_bytes(kpts, int(x["numMatchesIn"]), int(x["numMatchesValid"]), hyp) for x in st]))
        dom = max(kms, key=kms.get)
        ach = bpp * P / (kms[dom] * 1e-3) / 1e9
        # HBM bytes of the dominant step from the rocprofv3 PMC passes of `bench.py --preset stress --error-version E`
        # (profiles/run_stress_profiles.sh -> profiles/pmc_traffic.json), per launch chain step like `achieved`
        traffic, traffic_src = None, None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            e = t.get("%dx%dxH%dxE%dxfixedx%sx%s" % (frames, kpts, hyp, ev, "mfma" if c0.get_option("matcher_used") == 1 else "valu",
                                                      "fast" if c0.get_option("score") >= 1 else "exact"), {})
            traffic, traffic_src = e.get(dom), e.get("_source")
        except (OSError, ValueError):
            pass
        out["stress/E%d" % ev] = {
            "workload": "BASELINE configs[4]: %d pairs x %d kpts, H = %d fixed, errorVersion %d, one launch chain" % (P, kpts, hyp, ev),
            "ms_per_step": ms, "pairs_per_s": P / (ms * 1e-3), "steps": n, "kernel_ms": kms, "kernel_ms_sum": sum(kms.values()),
            "mean_valid_matches": float(st["numMatchesValid"].mean()), "mean_inliers": float(st["numInliers"].mean()),
            "accepted_pairs": int(st["accepted"].sum()),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_pair": bpp, "pairs_per_launch": P,
                         "avg_launch_ms": kms[dom], "traffic": traffic, "traffic_source": traffic_src,
                         "whole_call_frac": bpp * P / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        if not args.no_cpu_baseline:
            out["stress/E%d" % ev]["cpu"] = stress_cpu(seq, prm, cfg, hyp, st)
        if ctxs is not None and len(ctxs) >= 2 and P >= 4:
            # the same 8 pairs submitted like the timed region: two unequal chains (3 + 5 pairs) that are never joined
            b2 = [0, max(1, int(P * 0.45)), P]
            nq = len(ctxs) if queue is not None else 2
            pbq = [pb] + [PairBatchDevice(seq["pairs"], fs.max_kpts, device=str(dev)) for _ in range(nq - 1)]
            turn = [0]

            def chains_step():
                if queue is not None:   # like the timed region: whole batches to the library's chains in turn, an output block each
                    run_pairs_queue(queue, prm, cfg, TUM_FR1_K, fs, pbq[turn[0] % nq])
                    turn[0] += 1
                else:
                    run_pairs_split(ctxs[:2], chains[:2], prm, EST_FIXED, hyp, cfg.seed, TUM_FR1_K, fs, pb, bounds=b2, join=False)

            for _ in range(3):
                chains_step()
            torch.cuda.synchronize(dev)
            tc = time.perf_counter()
            for _ in range(n):
                chains_step()
            torch.cuda.synchronize(dev)
            cms = (time.perf_counter() - tc) / n * 1e3
            out["stress/E%d" % ev]["chains"] = {"streams": nq, "submit": ("queue: whole batches in turn" if queue is not None else "split %r" % (b2,)), "ms_per_step": cms, "pairs_per_s": P / (cms * 1e-3), "steps": n}
    return out


def stress_cpu(seq, prm, cfg, hyp, st):
    """The oracle on the stress configuration's own frames (the checker timed as a baseline, never the product): all 8 pairs on
    min(cores, 8) threads (the oracle's parallelism is over pairs), and ONE pair on one thread when that fits the budget --
    about 5 s of wall per error version (H x matches evaluations at ~10 ns each)."""
    from oracle import oracle_py as po
    from putslam_amd._abi import TUM_FR1_K
    cores = usable_cores()
    pairs = seq["pairs"]
    P = len(pairs)
    po.set_matcher_simd(True)
    threads = max(1, min(cores, P))
    t0 = time.perf_counter()
    po.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], pairs, threads=threads)
    wall = time.perf_counter() - t0
    out = {"value": P / wall, "unit": "frame-pairs/s", "cores": threads, "kind": "port", "matcher": po.matcher_simd_kind(),
           "sample": "the leg's %d pairs once, OpenMP over pairs on %d threads, %.2f s wall" % (P, threads, wall)}
    est_one = wall * threads / P * (1.0 if threads <= P else 1.0)          # a pair's core-seconds in the run above
    if est_one <= 2.5:
        t1 = time.perf_counter()
        po.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], pairs[:1], threads=1)
        out["single_thread_value"] = 1.0 / (time.perf_counter() - t1)
        out["single_thread_sample"] = "pair 0 on one thread, measured"
    else:
        out["single_thread_value"] = 1.0 / est_one
        out["single_thread_sample"] = ("derived: a pair's core-seconds in the all-pairs run (one pair per thread); a measured pass "
                                       "would take %.1f s of the line's budget" % est_one)
    return out


def usable_cores():
    """Threads this process may really use: CPU affinity capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args, seq, prm, cfg, est):
    """The oracle (CPU restatement of the reference algorithm) timed on this host's cores, on a bounded
    sample of the same workload.  It is the checker being timed as a baseline, never the product."""
    from oracle import oracle_py as po
    from putslam_amd._abi import EST_RANSAC, TUM_FR1_K, make_config
    cores = usable_cores()
    pairs = seq["pairs"]

    def run(cfg_, n, threads):
        t0 = time.perf_counter()
        po.vo_pairs(prm, cfg_, TUM_FR1_K, seq["desc"], seq["pts"], seq["nkpts"], pairs[:n], threads=threads)
        return time.perf_counter() - t0

    # the matcher half runs the SIMD popcount sweep (what OpenCV's vectorised normHamming amounts to): the generous figure
    po.set_matcher_simd(True)
    n0 = min(len(pairs), max(cores, 2))
    t = run(cfg, n0, cores)                                   # calibration (also warms the pages)
    passes = 5
    n = int(min(len(pairs), max(n0, n0 * (args.cpu_seconds / passes) / max(t, 1e-3))))
    times = sorted(run(cfg, n, cores) for _ in range(passes))
    med = times[passes // 2]
    out = {"cpu_baseline": {"value": n / med, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                            "passes": passes, "pass_seconds": times, "matcher": po.matcher_simd_kind(),
                            "sample": f"first {n} of {len(pairs)} pairs of the same sequence, {passes} passes, median; same "
                                      f"H/errorVersion/estimator as the GPU run, OpenMP over pairs on {cores} threads, "
                                      f"{sum(times):.1f} s wall in all"}}
    po.set_matcher_simd(False)
    out["cpu_baseline"]["scalar_matcher_value"] = n / run(cfg, n, cores)     # one pass with the scalar popcnt sweep
    po.set_matcher_simd(True)
    # what the reference itself would do: sequential adaptive schedule, <= 487 iterations (RANSAC.cpp:30,450-453)
    cfg_ref, _ = make_config(EST_RANSAC, 487, seed=cfg.seed)
    n2 = min(len(pairs), max(n, 4 * cores))
    t2 = run(cfg_ref, n2, cores)
    t1 = run(cfg_ref, min(n2, 16), 1)
    out["cpu_reference_schedule"] = {
        "value": n2 / t2, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
        "single_thread_value": min(n2, 16) / t1, "matcher": po.matcher_simd_kind(),
        "sample": f"{n2} pairs, adaptive RANSAC schedule of the reference (<=487 iterations), all cores; "
                  f"single_thread_value on {min(n2, 16)} pairs"}
    return out


if __name__ == "__main__":
    main()
