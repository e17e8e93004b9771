#!/usr/bin/env python3
"""Latency of the streaming call ps_vo_stream_push (one frame in, one pose out; PCIe and synchronisation included)
with and without the captured hipGraph (PUTSLAM_HIP_NO_GRAPH=1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from putslam_amd import api, synth  # noqa: E402
from putslam_amd._abi import EST_RANSAC, TUM_FR1_K, default_ransac_params, make_config  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    ctx = api.Context(0)
    seq = synth.make_sequence(40, n, config=3, index=5)
    for mode in (1, 0):
        prm = default_ransac_params(mode)
        st = api.VoStream(ctx, n)
        ts = []
        for rep in range(6):
            for f in range(40):
                cfg, _ = make_config(EST_RANSAC, 487, seed=f + 1)
                t0 = time.perf_counter()
                st.push(prm, cfg, TUM_FR1_K, seq["desc"][f], seq["pts"][f])
                ts.append(time.perf_counter() - t0)
        st.close()
        ts = np.array(ts[40:]) * 1e3
        print(f"kpts={n} errorVersion={mode} graph={'off' if os.environ.get('PUTSLAM_HIP_NO_GRAPH') == '1' else 'on'}: "
              f"median {np.median(ts):.3f} ms  p10 {np.percentile(ts, 10):.3f}  p90 {np.percentile(ts, 90):.3f}  "
              f"-> {1e3 / np.median(ts):.0f} pairs/s")


if __name__ == "__main__":
    main()
