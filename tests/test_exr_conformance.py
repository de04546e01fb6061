"""OpenEXR conformance of the shading-cache files, checked WITHOUT iris_amd.utils.exr's reader: this file carries its own
byte-level parser written from the OpenEXR file-layout document (magic / version field, attribute list, chlist, offset table,
scanline blocks, ZIP = zlib + byte-delta predictor + even/odd de-interleave), so that a file accepted here is one OpenEXR /
`cv2.imread(path, -1)` accepts -- which is how the reference re-reads the 13 maps per view
(utils/dataset/scannetpp/dataset.py:359-372, utils/dataset/synthetic_ldr.py:58-64; written at bake_shading.py:131,202-203 as
`cv2.imwrite(path, img[:, :, [2, 1, 0]])`: float32, channels B, G, R)."""
import struct
import zlib

import numpy as np
import pytest

REQUIRED = {  # attribute name -> (type name, payload size in bytes or None)   [OpenEXR file layout: "Header attributes (all files)"]
    "channels": ("chlist", None), "compression": ("compression", 1), "dataWindow": ("box2i", 16), "displayWindow": ("box2i", 16),
    "lineOrder": ("lineOrder", 1), "pixelAspectRatio": ("float", 4), "screenWindowCenter": ("v2f", 8), "screenWindowWidth": ("float", 4)}
LINES_PER_BLOCK = {0: 1, 2: 1, 3: 16}     # NO_COMPRESSION, ZIPS, ZIP


def _cstr(buf, pos):
    end = pos
    while buf[end] != 0:
        end += 1
    assert 1 <= end - pos <= 31, "attribute / channel names are 1..31 bytes unless the long-names flag is set"
    return buf[pos:end].decode("ascii"), end + 1


def parse_exr(buf):
    """Independent scanline-EXR parser -> dict(width, height, channels {name: (H,W) float32}, compression); asserts the layout rules."""
    assert buf[:4] == bytes([0x76, 0x2F, 0x31, 0x01]), "magic number"
    (ver,) = struct.unpack_from("<I", buf, 4)
    assert ver & 0xFF == 2, "file format version 2"
    assert ver & 0x200 == 0 and ver & 0x400 == 0 and ver & 0x800 == 0 and ver & 0x1000 == 0, "single-part scanline file: no tiled / long-name / deep / multipart flag"
    assert ver >> 13 == 0, "reserved version bits must be zero"
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos); pos += 4
        assert size >= 0 and name not in attrs
        attrs[name] = (typ, buf[pos:pos + size]); pos += size
    pos += 1
    for name, (typ, size) in REQUIRED.items():
        assert name in attrs, f"required attribute {name} missing"
        assert attrs[name][0] == typ, f"{name}: type {attrs[name][0]} != {typ}"
        if size is not None:
            assert len(attrs[name][1]) == size, f"{name}: {len(attrs[name][1])} bytes"
    # chlist: name\0, int pixelType, uchar pLinear, 3 reserved bytes, int xSampling, int ySampling; terminated by \0; sorted by name
    cb, p, chans = attrs["channels"][1], 0, []
    while cb[p] != 0:
        n, p = _cstr(cb, p)
        ptype, plinear, r0, r1, r2, xs, ys = struct.unpack_from("<iBBBBii", cb, p); p += 16
        assert ptype in (0, 1, 2) and plinear in (0, 1) and (r0, r1, r2) == (0, 0, 0) and xs == 1 and ys == 1
        chans.append((n, ptype))
    assert p + 1 == len(cb), "chlist must end with a single null byte"
    assert [n for n, _ in chans] == sorted(n for n, _ in chans), "channels are stored in alphabetical order"
    comp = attrs["compression"][1][0]
    assert comp in LINES_PER_BLOCK
    xmin, ymin, xmax, ymax = struct.unpack("<iiii", attrs["dataWindow"][1])
    assert struct.unpack("<iiii", attrs["displayWindow"][1]) == (xmin, ymin, xmax, ymax)
    assert attrs["lineOrder"][1][0] == 0, "INCREASING_Y"
    assert struct.unpack("<f", attrs["pixelAspectRatio"][1])[0] == 1.0 and struct.unpack("<f", attrs["screenWindowWidth"][1])[0] == 1.0
    W, H = xmax - xmin + 1, ymax - ymin + 1
    lines = LINES_PER_BLOCK[comp]
    n_blocks = (H + lines - 1) // lines
    offs = struct.unpack_from("<%dQ" % n_blocks, buf, pos)
    assert offs[0] == pos + 8 * n_blocks, "first block directly behind the offset table"
    assert all(b > a for a, b in zip(offs, offs[1:])), "INCREASING_Y files have a monotone offset table"
    psize = {0: 4, 1: 2, 2: 4}
    row_bytes = sum(psize[t] for _, t in chans) * W
    planes = {n: np.zeros((H, W), np.float32) for n, _ in chans}
    end = offs[0]
    for i, o in enumerate(offs):
        assert o == end, "blocks are contiguous"
        y, size = struct.unpack_from("<ii", buf, o)
        assert y == ymin + i * lines, "block i starts at scanline ymin + i * linesPerBlock"
        nl = min(lines, H - i * lines)
        raw_size = row_bytes * nl
        assert 0 < size <= raw_size, "a block is stored compressed only when that is smaller"
        data = buf[o + 8:o + 8 + size]
        end = o + 8 + size
        if comp != 0 and size < raw_size:
            t = zlib.decompress(data)
            assert len(t) == raw_size
            # predictor: t[i] = t[i-1] + d[i] - 128 (mod 256)
            d = bytearray(t)
            for k in range(1, raw_size):
                d[k] = (d[k - 1] + d[k] - 128) & 0xFF
            # interleave: first half -> even bytes, second half -> odd bytes
            half = (raw_size + 1) // 2
            out = bytearray(raw_size)
            out[0::2] = d[:half]
            out[1::2] = d[half:]
            data = bytes(out)
        q = 0
        for ly in range(nl):
            for n, t in chans:
                dt = {0: "<u4", 1: "<f2", 2: "<f4"}[t]
                planes[n][i * lines + ly] = np.frombuffer(data, dt, W, q).astype(np.float32)
                q += psize[t] * W
        assert q == raw_size
    assert end == len(buf), "no trailing bytes behind the last block"
    return {"width": W, "height": H, "channels": planes, "types": dict(chans), "compression": comp}


@pytest.mark.parametrize("comp", ["none", "zips", "zip"])
@pytest.mark.parametrize("hw", [(37, 53), (16, 16), (1, 1), (33, 7)])
def test_written_files_follow_the_openexr_layout(tmp_path, comp, hw):
    from iris_amd.utils import exr
    rng = np.random.default_rng(hw[0] * 131 + hw[1])
    img = (rng.random((*hw, 3)) * 9).astype(np.float32)
    img[0, 0] = [0.25, 1e-30, 65504.0]
    if comp != "none" and hw == (37, 53):
        img[5:30] = 0.5                      # long constant runs: the compressed path is certainly taken
    path = str(tmp_path / "m.exr")
    exr.write_exr(path, img, comp)
    f = parse_exr(open(path, "rb").read())
    assert (f["height"], f["width"]) == hw
    assert f["types"] == {"B": 2, "G": 2, "R": 2}, "three FLOAT channels named B, G, R"
    # cv2.imread(path, -1) returns the channels as (B, G, R) planes; the reference then takes [..., [2, 1, 0]] -> R, G, B
    np.testing.assert_array_equal(np.stack([f["channels"]["R"], f["channels"]["G"], f["channels"]["B"]], -1), img)


def test_reader_accepts_a_foreign_writer(tmp_path):
    """The other direction: a file assembled here byte by byte (HALF + FLOAT channels, ZIP, attributes in a different order, an
    extra attribute) is read correctly by iris_amd.utils.exr.read_exr -- i.e. the loader does not depend on its own writer's habits."""
    from iris_amd.utils import exr
    H, W = 19, 11
    rng = np.random.default_rng(3)
    img = rng.random((H, W, 3)).astype(np.float32)
    img16 = img.astype(np.float16).astype(np.float32)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    ch = b"".join(n + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1) for n, t in ((b"B", 1), (b"G", 2), (b"R", 1))) + b"\0"
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    hdr = bytes([0x76, 0x2F, 0x31, 0x01]) + struct.pack("<I", 2)
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + attr("owner", "string", b"test") + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("displayWindow", "box2i", box) + attr("dataWindow", "box2i", box) + attr("compression", "compression", bytes([3]))
    hdr += attr("channels", "chlist", ch) + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + b"\0"
    blocks = []
    for y0 in range(0, H, 16):
        raw = b""
        for y in range(y0, min(H, y0 + 16)):
            raw += img[y, :, 2].astype("<f2").tobytes() + img[y, :, 1].astype("<f4").tobytes() + img[y, :, 0].astype("<f2").tobytes()
        n = len(raw)
        t = raw[0::2] + raw[1::2]
        d = bytearray(t)
        for k in range(n - 1, 0, -1):
            d[k] = (t[k] - t[k - 1] + 128) & 0xFF
        z = zlib.compress(bytes(d))
        data = z if len(z) < n else raw
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    off, table = len(hdr) + 8 * len(blocks), b""
    for b in blocks:
        table += struct.pack("<Q", off); off += len(b)
    path = str(tmp_path / "foreign.exr")
    open(path, "wb").write(hdr + table + b"".join(blocks))
    got = exr.read_exr(path)
    np.testing.assert_array_equal(got[..., 0], img16[..., 0])
    np.testing.assert_array_equal(got[..., 1], img[..., 1])
    np.testing.assert_array_equal(got[..., 2], img16[..., 2])
