"""Independent reader of scanline OpenEXR files for the tests -- written from the published file layout
(https://openexr.com/en/latest/OpenEXRFileLayout.html), sharing no code with the host's writer / reader
(nori_amd/csrc/host/bitmap.cpp), so that what the host writes is checked by a second implementation:
magic and version field, the attribute list with its required attributes, the scan line offset table (entries must point
exactly at their blocks), and the pixel data of NONE / ZIPS / ZIP compressed files with HALF, FLOAT or UINT channels."""
import struct
import zlib

import numpy as np

REQUIRED = ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth")
PIXEL_TYPES = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
LINES_PER_BLOCK = {0: 1, 2: 1, 3: 16}      # NO_COMPRESSION, ZIPS_COMPRESSION, ZIP_COMPRESSION


def _cstr(b, i):
    e = b.index(b"\0", i)
    return b[i:e].decode("latin-1"), e + 1


def _unzip(block, expect):
    raw = zlib.decompress(block)
    assert len(raw) == expect
    a = np.frombuffer(raw, np.uint8).astype(np.int32)
    a = np.cumsum(a - np.concatenate(([0], np.full(len(a) - 1, 128)))) % 256      # predictor: d[i] = d[i-1] + t[i] - 128
    a = a.astype(np.uint8)
    half = (len(a) + 1) // 2
    out = np.empty(len(a), np.uint8)
    out[0::2] = a[:half]; out[1::2] = a[half:]                                     # de-interleave the two halves
    return out.tobytes()


def read_exr(path):
    """-> (header dict, {channel name: (H, W) array}).  Asserts the structural rules on the way."""
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", b, 0)
    assert magic == 20000630, "magic number 0x762f3101"
    assert version & 0xff == 2, "file format version 2"
    assert version & ~0xff & ~0x400 == 0, "single-part scan line file (no tiles, deep data or multiple parts)"
    long_names = bool(version & 0x400)
    i, hdr = 8, {}
    while True:
        name, i = _cstr(b, i)
        if not name:
            break
        assert len(name) <= (255 if long_names else 31)
        typ, i = _cstr(b, i)
        (size,) = struct.unpack_from("<i", b, i); i += 4
        val = b[i:i + size]; i += size
        assert len(val) == size and name not in hdr
        hdr[name] = (typ, val)
    for k in REQUIRED:
        assert k in hdr, f"required attribute {k}"
    out = {}
    for k, (typ, val) in hdr.items():
        if typ == "chlist":
            chans, j = [], 0
            while val[j] != 0:
                cname, j = _cstr(val, j)
                ptype, plinear, xs, ys = struct.unpack_from("<iB3xii", val, j); j += 16
                chans.append((cname, ptype, xs, ys))
            assert j + 1 == len(val)
            assert [c[0] for c in chans] == sorted(c[0] for c in chans), "channels are stored in alphabetical order"
            out[k] = chans
        elif typ == "box2i": out[k] = struct.unpack("<4i", val)
        elif typ in ("compression", "lineOrder"): out[k] = val[0]; assert len(val) == 1
        elif typ == "float": out[k] = struct.unpack("<f", val)[0]
        elif typ == "v2f": out[k] = struct.unpack("<2f", val)
        elif typ == "string": out[k] = val.decode("latin-1")
        else: out[k] = val
    x0, y0, x1, y1 = out["dataWindow"]
    W, H = x1 - x0 + 1, y1 - y0 + 1
    assert W > 0 and H > 0
    comp = out["compression"]
    assert comp in LINES_PER_BLOCK, f"compression {comp} not handled by this reader"
    lpb = LINES_PER_BLOCK[comp]
    n_blocks = (H + lpb - 1) // lpb
    offsets = struct.unpack_from(f"<{n_blocks}Q", b, i); i += 8 * n_blocks
    chans = out["channels"]
    assert all(xs == 1 and ys == 1 for _, _, xs, ys in chans)
    line_bytes = sum(PIXEL_TYPES[p].itemsize for _, p, _, _ in chans) * W
    planes = {c[0]: np.zeros((H, W), PIXEL_TYPES[c[1]]) for c in chans}
    expect_at = i
    order = range(n_blocks) if out["lineOrder"] == 0 else reversed(range(n_blocks))
    for blk in order:
        off = offsets[blk]
        if out["lineOrder"] in (0, 1):
            assert off == expect_at, "offset table entries point at consecutive blocks"
        y, size = struct.unpack_from("<ii", b, off)
        assert y == y0 + blk * lpb
        lines = min(lpb, H - blk * lpb)
        data = b[off + 8:off + 8 + size]
        assert len(data) == size
        expect_at = off + 8 + size
        if comp != 0 and size < lines * line_bytes:
            data = _unzip(data, lines * line_bytes)
        assert len(data) == lines * line_bytes
        p = 0
        for ln in range(lines):
            for cname, ptype, _, _ in chans:      # within a scan line: channel after channel
                dt = PIXEL_TYPES[ptype]
                planes[cname][blk * lpb + ln] = np.frombuffer(data, dt, W, p); p += dt.itemsize * W
    assert expect_at == len(b), "nothing after the last block"
    return out, planes


def read_exr_rgb(path):
    hdr, planes = read_exr(path)
    return hdr, np.stack([planes["R"], planes["G"], planes["B"]], axis=-1).astype(np.float32)
