"""Minimal OpenEXR (scanline, single part) writer / reader for the shading cache files.

The reference writes its 13 maps per view with ``cv2.imwrite(path, img[:, :, [2, 1, 0]])`` (bake_shading.py:131,202-203): float32
EXR files whose channels are named B, G, R and hold blue, green, red, and reads them back with
``cv2.imread(path, -1)[..., [2, 1, 0]]`` (utils/dataset/synthetic_ldr.py:58-64).  This module produces / consumes the same
files without OpenCV: channels R, G, B (stored alphabetically B, G, R as the format requires), FLOAT pixels, NO_COMPRESSION or
ZIP on write; NONE / ZIPS / ZIP, FLOAT / HALF on read.  Host-side I/O, not part of the hot path.
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_COMP_NONE, _COMP_ZIPS, _COMP_ZIP = 0, 2, 3
_LINES = {_COMP_NONE: 1, _COMP_ZIPS: 1, _COMP_ZIP: 16}


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


def _zip_compress(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int16)       # reorder: even bytes then odd bytes
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128                                    # delta predictor
    out = zlib.compress((d & 0xFF).astype(np.uint8).tobytes(), 6)
    return out if len(out) < len(raw) else raw


def _zip_decompress(buf, raw_size):
    if len(buf) == raw_size:
        return buf
    d = np.frombuffer(zlib.decompress(buf), np.uint8).astype(np.int64)
    d[1:] -= 128
    t = (np.cumsum(d) & 0xFF).astype(np.uint8)
    half = (raw_size + 1) // 2
    out = np.empty(raw_size, np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def write_exr(path, rgb, compression="none"):
    """rgb: (H,W,3) float array in R,G,B order -> float32 EXR with channels B,G,R."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    assert rgb.ndim == 3 and rgb.shape[2] == 3, "expected (H,W,3)"
    H, W, _ = rgb.shape
    comp = {"none": _COMP_NONE, "zips": _COMP_ZIPS, "zip": _COMP_ZIP}[compression]
    chl = b"".join(n + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<ii", _MAGIC, 2)
    hdr += _attr("channels", "chlist", chl) + _attr("compression", "compression", struct.pack("<B", comp))
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    lines = _LINES[comp]
    planes = rgb[:, :, [2, 1, 0]].transpose(0, 2, 1)                # (H, [B,G,R], W): per scanline, channel-contiguous
    chunks = []
    for y0 in range(0, H, lines):
        raw = planes[y0:y0 + lines].tobytes()
        data = raw if comp == _COMP_NONE else _zip_compress(raw)
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    off = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as fh:
        fh.write(hdr + table + b"".join(chunks))


def read_exr_header(path):
    with open(path, "rb") as fh:
        buf = fh.read(1 << 16)
    return _parse_header(buf)[0]


def _parse_header(buf):
    magic, version = struct.unpack_from("<ii", buf, 0)
    assert magic == _MAGIC, "not an OpenEXR file"
    assert (version & 0x1E00) == 0, "tiled / deep / multi-part EXR files are not supported"
    pos, attrs = 8, {}
    while buf[pos] != 0:
        e = buf.index(b"\0", pos); name = buf[pos:e].decode(); pos = e + 1
        e = buf.index(b"\0", pos); typ = buf[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", buf, pos); pos += 4
        attrs[name] = (typ, buf[pos:pos + size]); pos += size
    pos += 1
    chans, cb, p = [], attrs["channels"][1], 0
    while cb[p] != 0:
        e = cb.index(b"\0", p); n = cb[p:e].decode(); p = e + 1
        ptype, _, _, _, _, xs, ys = struct.unpack_from("<iBBBBii", cb, p); p += 16
        assert xs == 1 and ys == 1, "sub-sampled channels are not supported"
        chans.append((n, ptype))
    xmin, ymin, xmax, ymax = struct.unpack("<iiii", attrs["dataWindow"][1])
    info = {"channels": chans, "compression": attrs["compression"][1][0], "width": xmax - xmin + 1, "height": ymax - ymin + 1,
            "ymin": ymin, "line_order": attrs.get("lineOrder", ("", b"\0"))[1][0]}
    return info, pos


def read_exr(path):
    """-> (H,W,3) float32 in R,G,B order (what cv2.imread(path,-1)[...,[2,1,0]] returns for these files)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    info, pos = _parse_header(buf)
    H, W, comp = info["height"], info["width"], info["compression"]
    assert comp in _LINES, f"EXR compression {comp} is not supported (NONE, ZIPS, ZIP only)"
    lines = _LINES[comp]
    n_chunks = (H + lines - 1) // lines
    offs = struct.unpack_from("<%dQ" % n_chunks, buf, pos)
    sizes = {0: 4, 1: 2, 2: 4}
    dts = {0: "<u4", 1: "<f2", 2: "<f4"}
    row_bytes = sum(sizes[t] for _, t in info["channels"]) * W
    out = {n: np.zeros((H, W), np.float32) for n, _ in info["channels"]}
    for o in offs:
        y, sz = struct.unpack_from("<ii", buf, o)
        y0 = y - info["ymin"]
        nl = min(lines, H - y0)
        raw = buf[o + 8:o + 8 + sz]
        if comp != _COMP_NONE:
            raw = _zip_decompress(raw, row_bytes * nl)
        p = 0
        for ly in range(nl):
            for n, t in info["channels"]:
                out[n][y0 + ly] = np.frombuffer(raw, dts[t], W, p).astype(np.float32)
                p += sizes[t] * W
    if all(k in out for k in ("R", "G", "B")):
        return np.stack([out["R"], out["G"], out["B"]], -1)
    return np.stack([out[n] for n, _ in info["channels"]], -1)
