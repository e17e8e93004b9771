"""On-disk input of the path: the TUM-style sequence directory PUTSLAM's FileGrabber plays.

Mirrors `src/Grabber/fileGrabber.cpp:25-160` (non-real-time playback):
  * `<fullPath>/matched`       one line per frame: "<rgb timestamp> <depth timestamp>"; the frame's timestamp is the
                               mean of the two (`:82-86`); a line shorter than 3 characters ends the sequence (`:78`)
  * `<fullPath>/rgb_%05d.png`  colour image of frame fileNo (`:127-133`, cv::imread COLOR -> 8-bit, 3 channels, BGR)
  * `<fullPath>/depth_%05d.png` 16-bit depth image (`:136-144`, cv::imread ANYDEPTH); metres = value / depthImageScale
  * `playEveryNth` skips lines AND file numbers together (`:63-69`), `maxNumberOfFrames` bounds the frames returned
    (`:49-51`).
Frames are queued like the reference's MODE_BUFFER grabber (`:150-156`) and handed out by `get_sensor_frame()`.

PNG decoding does not depend on OpenCV: `read_png` / `write_png` below implement the subset the dataset uses
(non-interlaced, 8-bit gray / RGB / RGBA and 16-bit gray, all five scanline filters) on zlib + numpy.  Pillow, when
importable, is only used by the tests as an independent check.

Host-side plumbing only: the arithmetic that consumes these images (`ps_keypoints2Dto3D`, `ps_remove_image_distortion`)
stays behind the C ABI.
"""
import collections
import dataclasses
import os
import struct
import zlib

import numpy as np

_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"


# ------------------------------------------------------------------------------------------------ PNG
def _unfilter(raw, height, stride, bpp):
    """Undo the per-scanline PNG filters (PNG spec section 9). raw: bytes of height x (1 + stride)."""
    out = np.zeros((height, stride), np.uint8)
    rows = np.frombuffer(raw, np.uint8).reshape(height, stride + 1)
    prev = np.zeros(stride, np.uint8)
    for y in range(height):
        ft = int(rows[y, 0])
        line = rows[y, 1:]
        if ft == 0:
            cur = line.copy()
        elif ft == 1:      # Sub: running sum per byte lane, modulo 256
            cur = line.reshape(-1, bpp).astype(np.uint32)
            cur = (np.cumsum(cur, axis=0) & 0xFF).astype(np.uint8).reshape(-1)
        elif ft == 2:      # Up
            cur = (line.astype(np.uint16) + prev).astype(np.uint8)
        elif ft in (3, 4):  # Average / Paeth: true recurrences along the row
            cur = np.zeros(stride, np.uint8)
            ln, pv, cu = line.tolist(), prev.tolist(), [0] * stride
            if ft == 3:
                for i in range(stride):
                    left = cu[i - bpp] if i >= bpp else 0
                    cu[i] = (ln[i] + ((left + pv[i]) >> 1)) & 0xFF
            else:
                for i in range(stride):
                    a = cu[i - bpp] if i >= bpp else 0
                    b = pv[i]
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cu[i] = (ln[i] + pred) & 0xFF
            cur[:] = cu
        else:
            raise ValueError(f"PNG: unknown filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    """Decode a PNG into a numpy array: HxW (gray, uint8 or uint16) or HxWxC (uint8, channels in file order = RGB[A])."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_MAGIC:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        (length,), ctype = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        if ctype == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
        pos += 12 + length
    if hdr is None:
        raise ValueError(f"{path}: missing IHDR")
    width, height, depth, ctype, _comp, _flt, interlace = hdr
    if interlace:
        raise ValueError(f"{path}: interlaced PNG is not supported")
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (8, 16):
        raise ValueError(f"{path}: unsupported PNG colour type {ctype} / bit depth {depth}")
    bpp = channels * depth // 8
    stride = width * bpp
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != height * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    px = _unfilter(raw, height, stride, bpp)
    if depth == 16:
        img = px.reshape(height, width, channels, 2)
        img = (img[..., 0].astype(np.uint16) << 8) | img[..., 1]          # big-endian samples
    else:
        img = px.reshape(height, width, channels)
    return img[:, :, 0] if channels == 1 else img


def write_png(path, img):
    """Encode HxW uint8/uint16 (gray) or HxWx3 / HxWx4 uint8 as a non-interlaced PNG (filter 0 on every row)."""
    img = np.asarray(img)
    if img.ndim == 2:
        channels, ctype = 1, 0
    elif img.ndim == 3 and img.shape[2] in (3, 4):
        channels, ctype = img.shape[2], (2 if img.shape[2] == 3 else 6)
    else:
        raise ValueError("write_png: expected HxW or HxWx3/4")
    if img.dtype == np.uint16:
        if channels != 1:
            raise ValueError("write_png: 16-bit is supported for gray only")
        depth, body = 16, img.astype(">u2").tobytes()
    elif img.dtype == np.uint8:
        depth, body = 8, np.ascontiguousarray(img).tobytes()
    else:
        raise ValueError("write_png: dtype must be uint8 or uint16")
    h, w = img.shape[:2]
    stride = w * channels * depth // 8
    rows = np.frombuffer(body, np.uint8).reshape(h, stride)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_PNG_MAGIC + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ------------------------------------------------------------------------------------------------ grabber
@dataclasses.dataclass
class SensorFrame:
    """The fields of putslam::SensorFrame the path reads (include/putslam/Defs/putslam_defs.h)."""
    readId: int
    timestamp: float
    rgbImage: np.ndarray          # HxWx3 uint8, BGR like cv::imread
    depthImage: np.ndarray        # HxW uint16
    depthImageScale: float


class FileGrabber:
    """FileGrabber (fileGrabber.cpp:25-160), buffered mode, realTime = false."""

    def __init__(self, full_path, depth_image_scale=5000.0, play_every_nth=1, max_number_of_frames=2 ** 31 - 1):
        self.full_path = full_path if full_path.endswith(os.sep) else full_path + os.sep
        self.depth_image_scale = float(depth_image_scale)
        self.play_every_nth = int(play_every_nth)
        self.max_number_of_frames = int(max_number_of_frames)
        self.file_no = -1
        self.processing_file_counter = 0
        self._frames = collections.deque()
        with open(self.full_path + "matched") as f:
            self._lines = f.read().split("\n")       # std::getline view of the file
        self._line = 0

    def grab(self):
        if self._line >= len(self._lines):                                   # timestampFile.eof()
            return False
        if self.processing_file_counter >= self.max_number_of_frames:
            return False
        timestamp = 0.0
        for _ in range(max(self.play_every_nth, 0)):
            self.file_no += 1
            s = self._lines[self._line] if self._line < len(self._lines) else ""
            self._line += 1
            if len(s) < 3:
                return False
            sp = s.find(" ")
            t1 = _atof(s[:sp] if sp >= 0 else s)
            t2 = _atof(s[sp + 1:] if sp >= 0 else s)      # npos + 1 == 0: the whole line again
            timestamp = (t1 + t2) / 2
        tag = "%05d" % self.file_no
        rgb = read_png(self.full_path + "rgb_" + tag + ".png")
        if rgb.ndim == 2:
            rgb = np.repeat(rgb[:, :, None], 3, axis=2)                      # IMREAD_COLOR promotes gray
        rgb = np.ascontiguousarray(rgb[:, :, 2::-1])                         # RGB[A] -> BGR
        depth = read_png(self.full_path + "depth_" + tag + ".png")
        if depth.ndim != 2:
            raise ValueError("depth_" + tag + ".png: expected a single-channel image")
        self.processing_file_counter += 1
        self._frames.append(SensorFrame(self.file_no, timestamp, rgb, depth, self.depth_image_scale))
        return True

    def get_sensor_frame(self):
        """Oldest queued frame (Grabber::getSensorFrame pops the buffer)."""
        return self._frames.popleft()

    def __iter__(self):
        while self.grab():
            yield self.get_sensor_frame()


def _atof(s):
    """C atof: longest valid floating prefix, 0.0 when there is none."""
    s = s.lstrip()
    end, seen_digit, seen_dot, seen_exp, i = 0, False, False, False, 0
    if i < len(s) and s[i] in "+-":
        i += 1
    while i < len(s):
        c = s[i]
        if c.isdigit():
            seen_digit, end = True, i + 1
        elif c == "." and not seen_dot and not seen_exp:
            seen_dot = True
        elif c in "eE" and seen_digit and not seen_exp:
            seen_exp = True
            if i + 1 < len(s) and s[i + 1] in "+-":
                i += 1
        else:
            break
        i += 1
    try:
        return float(s[:end]) if end else 0.0
    except ValueError:
        return 0.0


def write_sequence(full_path, frames):
    """Writes a FileGrabber directory: frames = iterable of (rgb_timestamp, depth_timestamp, rgb HxWx3 BGR uint8,
    depth HxW uint16).  Used by the tests and demos to stage synthetic sequences in the dataset's format."""
    os.makedirs(full_path, exist_ok=True)
    lines = []
    for i, (t_rgb, t_depth, rgb, depth) in enumerate(frames):
        write_png(os.path.join(full_path, "rgb_%05d.png" % i), np.ascontiguousarray(rgb[:, :, ::-1]))
        write_png(os.path.join(full_path, "depth_%05d.png" % i), depth)
        lines.append("%.6f %.6f" % (t_rgb, t_depth))
    with open(os.path.join(full_path, "matched"), "w") as f:
        f.write("\n".join(lines) + "\n")
