"""Minimal OpenEXR (scanline, single part) writer / reader for the shading cache files.

The reference writes its 13 maps per view with ``cv2.imwrite(path, img[:, :, [2, 1, 0]])`` (bake_shading.py:131,202-203): float32
EXR files whose channels are named B, G, R and hold blue, green, red, and reads them back with
``cv2.imread(path, -1)[..., [2, 1, 0]]`` (utils/dataset/synthetic_ldr.py:58-64).  This module produces / consumes the same
files without OpenCV: channels R, G, B (stored alphabetically B, G, R as the format requires), FLOAT pixels, NO_COMPRESSION or
ZIP on write; NONE / ZIPS / ZIP, FLOAT / HALF on read.  Host-side I/O, not part of the hot path.
"""
import struct
import os
import zlib

import numpy as np

_MAGIC = 20000630
_COMP_NONE, _COMP_ZIPS, _COMP_ZIP = 0, 2, 3
_LINES = {_COMP_NONE: 1, _COMP_ZIPS: 1, _COMP_ZIP: 16}


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


_ZLEVEL = 4      # OpenEXR 3's default deflate level (Monte-Carlo float data barely compresses at any level)


def _predict(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]])                          # reorder: even bytes then odd bytes
    d = np.empty_like(t)
    d[0] = t[0]
    np.subtract(t[1:], t[:-1], out=d[1:])                           # delta predictor, modulo 256
    d[1:] += 128
    return d.tobytes()


def _unpredict(pred):
    d = np.frombuffer(pred, np.uint8).copy()
    d[1:] -= 128                                                    # uint8 arithmetic wraps: modulo 256, as the format defines it
    t = np.cumsum(d, dtype=np.uint8)
    half = (len(pred) + 1) // 2
    out = np.empty(len(pred), np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _deflate_predicted(pred):
    """predicted block -> what the file stores: the deflate stream, or the RAW block when deflate does not shrink it (OpenEXR's rule)"""
    out = zlib.compress(pred, _ZLEVEL)
    return out if len(out) < len(pred) else _unpredict(pred)


def _zip_compress(raw):
    return _deflate_predicted(_predict(raw))


_chunk_pool = None


def chunk_pool():
    """Shared thread pool for the scanline blocks of the files being written (zlib releases the GIL on bytes objects): deflate of noisy f32
    data runs at ~30 MB/s per core and a 1080p view is 13 maps x 25 MB, so one thread per FILE would leave the writer several times slower
    than the bake on a many-core host."""
    global _chunk_pool
    if _chunk_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _chunk_pool = ThreadPoolExecutor(max_workers=max(1, min(64, os.cpu_count() or 4)), thread_name_prefix="exr")
    return _chunk_pool


def _zip_decompress(buf, raw_size):
    if len(buf) == raw_size:
        return buf
    return _unpredict(zlib.decompress(buf))


def _header(H, W, comp):
    chl = b"".join(n + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<ii", _MAGIC, 2)
    hdr += _attr("channels", "chlist", chl) + _attr("compression", "compression", struct.pack("<B", comp))
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    return hdr


def _write_chunks(path, hdr, chunks):
    off = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as fh:
        fh.write(hdr + table)
        for c in chunks:
            fh.write(c)


def write_exr(path, rgb, compression="none", pool=None):
    """rgb: (H,W,3) float array in R,G,B order -> float32 EXR with channels B,G,R.  pool: an executor the scanline blocks are compressed on
    (chunk_pool(); must not be the pool this call itself runs on)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    assert rgb.ndim == 3 and rgb.shape[2] == 3, "expected (H,W,3)"
    H, W, _ = rgb.shape
    comp = {"none": _COMP_NONE, "zips": _COMP_ZIPS, "zip": _COMP_ZIP}[compression]
    hdr = _header(H, W, comp)
    lines = _LINES[comp]
    planes = rgb[:, :, [2, 1, 0]].transpose(0, 2, 1)                # (H, [B,G,R], W): per scanline, channel-contiguous

    def encode(y0):
        raw = planes[y0:y0 + lines].tobytes()
        data = raw if comp == _COMP_NONE else _zip_compress(raw)
        return struct.pack("<ii", y0, len(data)) + data
    starts = range(0, H, lines)
    chunks = list(pool.map(encode, starts)) if (pool is not None and comp != _COMP_NONE) else [encode(y0) for y0 in starts]
    _write_chunks(path, hdr, chunks)


def scanline_blocks_torch(maps, compression):
    """Device-side half of the writer.  maps: (M,H,W,3) f32 tensor in R,G,B order (any device) -> (full, tail): uint8 tensors (M, n_full,
    block_bytes) and (M, tail_bytes) (tail_bytes may be 0) holding the bytes of every scanline block as write_exr produces them BEFORE
    deflate: channels B,G,R planar per scanline and, for ZIP / ZIPS, reordered (even bytes, odd bytes) and delta-predicted.  A handful of
    streaming tensor operations on the GPU instead of ~0.1 s of numpy per map under the GIL on the host."""
    import torch
    comp = {"none": _COMP_NONE, "zips": _COMP_ZIPS, "zip": _COMP_ZIP}[compression]
    M, H, W, _ = maps.shape
    lines = _LINES[comp]
    planes = maps.to(torch.float32).flip(-1).permute(0, 1, 3, 2).contiguous().view(torch.uint8).reshape(M, H, 3 * W * 4)    # (M, H, row bytes)

    def blocks(rows, n):                                            # (M, n * lines', row_bytes) -> (M, n, lines' * row_bytes) in file byte order
        b = rows.reshape(M, n, -1)
        if comp == _COMP_NONE:
            return b.contiguous()
        t = torch.cat([b[..., 0::2], b[..., 1::2]], dim=-1)
        d = t.clone()
        d[..., 1:] = t[..., 1:] - t[..., :-1] + 128                 # uint8 arithmetic wraps: modulo 256
        return d
    n_full = H // lines
    full = blocks(planes[:, :n_full * lines], n_full) if n_full else torch.empty(M, 0, lines * 3 * W * 4, dtype=torch.uint8, device=maps.device)
    rest = H - n_full * lines
    tail = blocks(planes[:, n_full * lines:], 1).reshape(M, -1) if rest else torch.empty(M, 0, dtype=torch.uint8, device=maps.device)
    return full, tail


def write_exr_blocks(path, H, W, compression, full, tail, pool=None):
    """Host-side half: full (n_full, block_bytes) / tail (tail_bytes,) uint8 arrays of ONE map from scanline_blocks_torch -> the same file
    write_exr writes.  Only deflate (on `pool`, GIL-free) and file I/O happen here."""
    comp = {"none": _COMP_NONE, "zips": _COMP_ZIPS, "zip": _COMP_ZIP}[compression]
    hdr = _header(H, W, comp)
    lines = _LINES[comp]
    parts = [full[i] for i in range(full.shape[0])] + ([tail] if tail.shape[0] else [])

    def encode(i):
        raw = parts[i].tobytes()
        data = raw if comp == _COMP_NONE else _deflate_predicted(raw)
        return struct.pack("<ii", i * lines, len(data)) + data
    idx = range(len(parts))
    chunks = list(pool.map(encode, idx)) if (pool is not None and comp != _COMP_NONE) else [encode(i) for i in idx]
    _write_chunks(path, hdr, chunks)


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
        types = {t for _, t in info["channels"]}
        if len(types) == 1:                                          # (the usual case) one pixel type: the block is an (nl, channels, W) array
            blk = np.frombuffer(raw, dts[types.pop()], nl * len(info["channels"]) * W).reshape(nl, len(info["channels"]), W)
            for ci, (n, _) in enumerate(info["channels"]):
                out[n][y0:y0 + nl] = blk[:, ci]
            continue
        p = 0
        for ly in range(nl):
            for n, t in info["channels"]:
                out[n][y0 + ly] = np.frombuffer(raw, dts[t], W, p).astype(np.float32)
                p += sizes[t] * W
    if all(k in out for k in ("R", "G", "B")):
        return np.stack([out["R"], out["G"], out["B"]], -1)
    return np.stack([out[n] for n, _ in info["channels"]], -1)
