#!/usr/bin/env python3
"""Does kernel 2's record traffic cost anything (counter traffic is 3.6 x the algorithmic bytes, kernel 2
writes 69 MB per step for 19 MB of match output)?  The records it writes are all read later (A - D by the value-exact code,
the reorder launch and kernel 4; F by every scoring stage): there is no smaller set to write.  So the A/B goes the other way: the
-DPS_STREAM_DIAG build writes every record a SECOND time into a shadow block nothing reads
(PUTSLAM_HIP_DIAG_SHADOW_RECORDS=1: + 112 B per depth-valid match, + 73 MB per 499-pair step).  If doubling the write traffic
moves neither the resident headline nor the streamed leg -- which shares L2 and fabric with 36 - 44 GB/s of SDMA uploads -- the
traffic is not what bounds them.  Alternating runs of bench.py on one box; usage: python profiles/scripts/r06_records_traffic_ab.py [turns]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
turns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = os.path.join(ROOT, "putslam_amd", "libputslam_hip_diag.so")
rows = {0: [], 1: []}
for t in range(turns):
    for shadow in (0, 1):
        env = dict(os.environ, PUTSLAM_HIP_LIB=lib, PUTSLAM_HIP_DIAG_SHADOW_RECORDS=str(shadow))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-stress", "--no-data-legs", "--no-latency",
                            "--no-native-legs"], env=env, capture_output=True, text=True, timeout=600)
        j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        om = j["other_modes"]
        row = (j["value"], j["single_chain"]["pairs_per_s"], om["streamed"]["pairs_per_s"], om["streamed/poses"]["pairs_per_s"],
               j["kernel_ms"].get("ps_crosscheck_prep"))
        rows[shadow].append(row)
        print("turn %d shadow %d: resident %.0f  single chain %.0f  streamed %.0f  streamed/poses %.0f  kernel 2 %.4f ms" % ((t, shadow) + row), flush=True)
for shadow in (0, 1):
    r = sorted(rows[shadow])
    med = [sorted(x[i] for x in rows[shadow])[len(r) // 2] for i in range(5)]
    print("median shadow %d: resident %.0f  single chain %.0f  streamed %.0f  streamed/poses %.0f  kernel 2 %.4f ms" % ((shadow,) + tuple(med)))
