"""N4 on-disk input: PNG subset codec and the FileGrabber mirror (fileGrabber.cpp:25-160)."""
import os

import numpy as np
import pytest

from putslam_amd import tum_io


def _images(seed, h=48, w=64):
    r = np.random.default_rng(seed)
    rgb = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # smooth depth with holes, like a Kinect frame (values 0 and 500..30000)
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (5000 + 40 * xx + 25 * yy + r.integers(0, 30, (h, w))).astype(np.uint16)
    depth[r.random((h, w)) < 0.05] = 0
    return rgb, depth


def test_png_roundtrip_own_codec(tmp_path):
    rgb, depth = _images(1)
    for name, img in (("c.png", rgb), ("d.png", depth), ("g.png", rgb[:, :, 0]),
                      ("a.png", np.concatenate([rgb, rgb[:, :, :1]], axis=2))):
        p = str(tmp_path / name)
        tum_io.write_png(p, img)
        back = tum_io.read_png(p)
        assert back.dtype == img.dtype and back.shape == img.shape
        assert np.array_equal(back, img)


def test_png_against_pillow(tmp_path):
    """Independent check in both directions; Pillow's encoder picks adaptive filters (Sub/Up/Average/Paeth)."""
    Image = pytest.importorskip("PIL.Image")
    rgb, depth = _images(2)
    smooth = (np.add.outer(np.arange(48), np.arange(64)) % 256).astype(np.uint8)
    p = str(tmp_path / "x.png")
    for img, mode in ((rgb, "RGB"), (depth, "I;16"), (smooth, "L")):
        tum_io.write_png(p, img)
        got = np.array(Image.open(p))
        assert np.array_equal(got.astype(img.dtype), img)
        Image.fromarray(img).save(p, optimize=True)
        assert np.array_equal(tum_io.read_png(p), img)


def test_png_all_filter_types(tmp_path):
    """Hand-filtered rows: every filter type must decode back to the same pixels."""
    import struct
    import zlib
    r = np.random.default_rng(3)
    h, w, bpp = 10, 17, 3
    img = r.integers(0, 256, (h, w * bpp), dtype=np.uint8)
    raw = bytearray()
    for y in range(h):
        ft = y % 5
        cur = img[y].astype(np.int32)
        prev = img[y - 1].astype(np.int32) if y else np.zeros(w * bpp, np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        raw.append(ft)
        raw += bytes((f & 0xFF).astype(np.uint8))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    p = str(tmp_path / "f.png")
    with open(p, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    assert np.array_equal(tum_io.read_png(p), img.reshape(h, w, bpp))


def test_png_rejects_bad_input(tmp_path):
    p = str(tmp_path / "bad.png")
    open(p, "wb").write(b"not a png at all")
    with pytest.raises(ValueError):
        tum_io.read_png(p)
    with pytest.raises(ValueError):
        tum_io.write_png(p, np.zeros((4, 4), np.float32))


def _stage(tmp_path, n):
    frames = []
    for i in range(n):
        rgb, depth = _images(10 + i)
        frames.append((1305031102.175304 + 0.033 * i, 1305031102.160407 + 0.033 * i, rgb, depth))
    d = str(tmp_path / "seq")
    tum_io.write_sequence(d, frames)
    return d, frames


def test_file_grabber_plays_sequence(tmp_path):
    d, frames = _stage(tmp_path, 5)
    g = tum_io.FileGrabber(d, depth_image_scale=5000.0)
    got = list(g)
    assert [f.readId for f in got] == [0, 1, 2, 3, 4]
    for f, (t1, t2, rgb, depth) in zip(got, frames):
        assert f.timestamp == (float("%.6f" % t1) + float("%.6f" % t2)) / 2      # mean of the two columns (:82-86)
        assert np.array_equal(f.rgbImage, rgb) and np.array_equal(f.depthImage, depth)
        assert f.depthImageScale == 5000.0
    assert g.grab() is False                                                        # end of file stays ended


def test_file_grabber_every_nth_and_max_frames(tmp_path):
    d, frames = _stage(tmp_path, 7)
    g = tum_io.FileGrabber(d, play_every_nth=3)
    ids = [f.readId for f in g]
    assert ids == [2, 5]            # lines and file numbers advance together; the 7th line has no partners left
    g = tum_io.FileGrabber(d, max_number_of_frames=3)
    assert [f.readId for f in g] == [0, 1, 2]
    g = tum_io.FileGrabber(d)
    assert g.grab() and g.grab()
    assert g.get_sensor_frame().readId == 0 and g.get_sensor_frame().readId == 1    # MODE_BUFFER: FIFO


def test_atof_prefix_semantics():
    assert tum_io._atof("1305031102.175304") == 1305031102.175304
    assert tum_io._atof("12.5abc") == 12.5
    assert tum_io._atof("abc") == 0.0
    assert tum_io._atof("  -3e2x") == -300.0


def test_grabbed_depth_backprojects_like_oracle(tmp_path):
    """FileGrabber frame -> RGBD::keypoints2Dto3D (oracle): zero-depth pixels give NaN-free zeros, others metres."""
    from oracle import oracle_py as orc
    from putslam_amd._abi import TUM_FR1_K
    d, frames = _stage(tmp_path, 1)
    fr = next(iter(tum_io.FileGrabber(d)))
    r = np.random.default_rng(0)
    xy = np.stack([r.uniform(0, 62.4, 200), r.uniform(0, 46.4, 200)], axis=1).astype(np.float32)
    pts = orc.keypoints2Dto3D(xy, fr.depthImage, TUM_FR1_K, fr.depthImageScale)
    ui, vi = np.rint(xy[:, 0]).astype(int), np.rint(xy[:, 1]).astype(int)
    z = fr.depthImage[vi, ui].astype(np.float64) / 5000.0
    assert np.allclose(pts[:, 2], z.astype(np.float32))
