"""Child process of tests/test_gpu_stream_async.py::test_failed_chunk_is_dropped_as_a_unit: the pipelined stream of a library
built with -DPS_STREAM_DIAG (PUTSLAM_HIP_LIB names it) whose N-th chunk fails -- before its batched call has queued anything
(PUTSLAM_HIP_STREAM_DIAG_FAIL_CHUNK) or behind it (..._FAIL_AFTER).  What the header promises: the failing push returns an
error, the chunk's frames are dropped as a unit, no place is lost, and the stream goes on in a new epoch whose pair numbering
(= hypothesis seeds) starts at 0.  Everything popped is compared with the oracle.  Exit code 0 = all of that held."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import oracle_py as po
    from putslam_amd import api, synth
    from putslam_amd._abi import EST_FIXED, REPROJECTION_ERROR, TUM_FR1_K, default_ransac_params, make_config
    from test_gpu_stream_async import _check_block
    assert "diag" in os.environ.get("PUTSLAM_HIP_LIB", "")
    fail_at = int(os.environ.get("PUTSLAM_HIP_STREAM_DIAG_FAIL_CHUNK") or os.environ["PUTSLAM_HIP_STREAM_DIAG_FAIL_AFTER"])
    B, F = 4, 40
    seq = synth.make_sequence(F, 500, config=3, index=606)
    prm = default_ransac_params(REPROJECTION_ERROR)
    cfg, _ = make_config(EST_FIXED, 512, seed=0xFA17)
    ctx = api.Context(0)
    st = api.VoStream(ctx, 500)
    st.configure_async(prm, cfg, TUM_FR1_K, chunk_frames=B, lanes=2)
    blocks, failed_at_frame = [], None
    f = 0
    while f < F:
        try:
            ok = st.push_async(seq["desc"][f], seq["pts"][f])
        except api.PsError as e:
            assert failed_at_frame is None, "only one chunk was to fail"
            assert "dropped" in str(e) and "new epoch" in str(e), str(e)
            failed_at_frame = f
            f += 1                                     # the frame went down with its chunk
            continue
        if not ok:                                     # PS_ERR_BUSY: flow control, nothing taken
            blk = st.pop_many(wait=True)
            assert blk is not None
            blocks.append(blk)
            continue
        f += 1
    st.flush()
    while True:
        blk = st.pop_many(wait=True)
        if blk is None:
            break
        blocks.append(blk)
    assert st.pending() == 0
    # the failing launch was chunk number `fail_at`: frames [B * fail_at, B * fail_at + B) are gone
    lost_lo, lost_hi = B * fail_at, B * fail_at + B
    assert failed_at_frame == lost_hi - 1, (failed_at_frame, lost_hi)
    # epoch 0: pairs 0 .. lost_lo - 2 of the sequence (frames 0 .. lost_lo - 1); epoch 1: the frames from lost_hi on, numbered from 0
    c0 = (po.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"][:lost_lo], seq["pts"][:lost_lo], seq["nkpts"][:lost_lo], seq["pairs"][:lost_lo - 1], threads=4)
          if lost_lo >= 2 else None)
    n1 = F - lost_hi
    c1 = po.vo_pairs(prm, cfg, TUM_FR1_K, seq["desc"][lost_hi:], seq["pts"][lost_hi:], seq["nkpts"][lost_hi:], seq["pairs"][:n1 - 1], threads=4)
    got = {0: 0, 1: 0}
    for blk in blocks:
        ep = blk["epoch"]
        assert ep in (0, 1) and blk["first_pair"] == got[ep], (ep, blk["first_pair"], got)
        _check_block(blk, c0 if ep == 0 else c1, got[ep])
        got[ep] += blk["count"]
    assert got == {0: max(lost_lo - 1, 0), 1: n1 - 1}, got
    # no place was lost: the pipeline still takes its full complement of chunks without a pop (two lanes -> six places)
    taken = 0
    for k in range(6 * B):
        if not st.push_async(seq["desc"][k % F], seq["pts"][k % F]):
            break
        taken += 1
    assert taken == 6 * B, taken
    st.close()
    ctx.close()
    print("ok: chunk %d failed, frames [%d, %d) dropped, %d + %d pairs equal the oracle" % (fail_at, lost_lo, lost_hi, got[0], got[1]))


if __name__ == "__main__":
    main()
