#!/usr/bin/env python3
"""Conformance-oracle harness (SURVEY.md §7 step 0, §8c "stand-in oracle").

The build container ships Pillow 12.2 whose libavif 1.4.1 links dav1d 1.5.3
(decoder) and libaom 3.13.2 (encoder).  This module

  * wraps raw AV1 OBUs (what `av1mi_encode_chunk` produces) in a minimal AVIF
    still-image container (`wrap_avif`),
  * decodes an AVIF with dav1d and returns the *exact* Y/U/V planes at native bit
    depth by calling libavif's public C API through ctypes (`decode_yuv`) -
    Pillow's own plugin only returns 8-bit RGB, which is not exact for 4:2:0,
  * encodes planes with libaom at a fixed CQ level for the reported-only CPU
    baseline (`libaom_encode_yuv`).

It is test/bench infrastructure: it is used to GENERATE the golden fixtures under
tests/golden/ (tools/make_golden.py) and, when present on the box, by bench.py's
reported-only cpu_baseline leg.  Tests never require it.
"""
import ctypes as C
import glob
import os
import struct

import numpy as np

_lib = None


def have_libavif():
    try:
        _load()
        return True
    except Exception:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    import PIL
    cands = glob.glob(os.path.join(os.path.dirname(PIL.__file__), "..", "pillow.libs", "libavif*.so*"))
    if not cands:
        raise RuntimeError("libavif (bundled with Pillow) not found")
    lib = C.CDLL(cands[0])
    lib.avifDecoderCreate.restype = C.c_void_p
    lib.avifDecoderDestroy.argtypes = [C.c_void_p]
    lib.avifImageCreateEmpty.restype = C.c_void_p
    lib.avifImageDestroy.argtypes = [C.c_void_p]
    lib.avifDecoderReadMemory.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    lib.avifDecoderReadMemory.restype = C.c_int
    lib.avifResultToString.argtypes = [C.c_int]
    lib.avifResultToString.restype = C.c_char_p
    lib.avifVersion.restype = C.c_char_p
    _lib = lib
    return lib


class _AvifImageHead(C.Structure):
    # leading fields of `struct avifImage` (libavif 1.x public header avif.h)
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32),
        ("yuvFormat", C.c_int), ("yuvRange", C.c_int), ("yuvChromaSamplePosition", C.c_int),
        ("yuvPlanes", C.c_void_p * 3), ("yuvRowBytes", C.c_uint32 * 3),
    ]


class _AvifImageColour(C.Structure):
    # `struct avifImage` up to the CICP fields (libavif 1.x avif.h: avifColorPrimaries & co. are uint16_t typedefs)
    _fields_ = _AvifImageHead._fields_ + [
        ("imageOwnsYUVPlanes", C.c_int), ("alphaPlane", C.c_void_p), ("alphaRowBytes", C.c_uint32), ("imageOwnsAlphaPlane", C.c_int),
        ("alphaPremultiplied", C.c_int), ("icc_data", C.c_void_p), ("icc_size", C.c_size_t),
        ("colorPrimaries", C.c_uint16), ("transferCharacteristics", C.c_uint16), ("matrixCoefficients", C.c_uint16),
    ]


def decode_colour(avif_bytes):
    """(color_primaries, transfer_characteristics, matrix_coefficients, full_range) as libavif reports them for a still WITHOUT a colr
    box: taken from the AV1 sequence header's color_config (its own parser: an independent reading of the header the oracle wrote)."""
    lib = _load()
    dec = lib.avifDecoderCreate()
    img = lib.avifImageCreateEmpty()
    try:
        buf = bytes(avif_bytes)
        r = lib.avifDecoderReadMemory(dec, img, buf, len(buf))
        if r != 0:
            raise RuntimeError("libavif/dav1d decode failed: %s" % lib.avifResultToString(r).decode())
        h = _AvifImageColour.from_address(img)
        return h.colorPrimaries, h.transferCharacteristics, h.matrixCoefficients, h.yuvRange
    finally:
        lib.avifImageDestroy(img)
        lib.avifDecoderDestroy(dec)


# ------------------------------------------------------------------ ISO-BMFF writer
def _box(kind, payload):
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _fullbox(kind, version, flags, payload):
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def av1c_bytes(depth=8, mono=False, seq_profile=0, seq_level_idx=31, ss=(1, 1)):
    b0 = 0x81
    b1 = (seq_profile << 5) | (seq_level_idx & 31)
    high = 1 if depth > 8 else 0
    twelve = 1 if depth == 12 else 0
    b2 = (0 << 7) | (high << 6) | (twelve << 5) | ((1 if mono else 0) << 4) | (ss[0] << 3) | (ss[1] << 2) | 0
    return bytes([b0, b1, b2, 0])


def wrap_avif(obus, width, height, depth=8, mono=False, seq_level_idx=31):
    """Minimal still-image AVIF around one temporal unit (layout verified against
    dav1d in SURVEY.md §B.5): ftyp, meta{hdlr pitm iloc iinf iprp{ipco{ispe pixi av1C} ipma}}, mdat."""
    ftyp = _box(b"ftyp", b"avif" + struct.pack(">I", 0) + b"avif" + b"mif1" + b"miaf")
    hdlr = _fullbox(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"pict") + b"\0" * 12 + b"\0")
    pitm = _fullbox(b"pitm", 0, 0, struct.pack(">H", 1))
    infe = _fullbox(b"infe", 2, 0, struct.pack(">HH4s", 1, 0, b"av01") + b"\0")
    iinf = _fullbox(b"iinf", 0, 0, struct.pack(">H", 1) + infe)
    ispe = _fullbox(b"ispe", 0, 0, struct.pack(">II", width, height))
    nch = 1 if mono else 3
    pixi = _fullbox(b"pixi", 0, 0, bytes([nch] + [depth] * nch))
    av1c = _box(b"av1C", av1c_bytes(depth, mono, 0, seq_level_idx))
    ipco = _box(b"ipco", ispe + pixi + av1c)
    ipma = _fullbox(b"ipma", 0, 0, struct.pack(">IHB", 1, 1, 3) + bytes([1, 2, 0x80 | 3]))
    iprp = _box(b"iprp", ipco + ipma)

    def meta_with(offset):
        iloc = _fullbox(b"iloc", 0, 0, bytes([0x44, 0x00]) + struct.pack(">HHHHII", 1, 1, 0, 1, offset, len(obus)))
        return _fullbox(b"meta", 0, 0, hdlr + pitm + iloc + iinf + iprp)

    meta = meta_with(0)
    offset = len(ftyp) + len(meta) + 8
    meta = meta_with(offset)
    return ftyp + meta + _box(b"mdat", bytes(obus))


def wrap_avis(samples, width, height, depth=8, sync=None, seq_level_idx=31):
    """Minimal AVIF image SEQUENCE ('avis' brand: moov/trak/stbl, one sample per temporal unit) so that
    dav1d (through libavif) decodes inter-coded frames in order.  `sync`: 1-based numbers of key samples."""
    n = len(samples)
    sync = sync if sync is not None else [1]
    ftyp = _box(b"ftyp", b"avis" + struct.pack(">I", 0) + b"avis" + b"msf1" + b"iso8" + b"mif1" + b"miaf" + b"MA1B")
    mat = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _fullbox(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 30, n) + struct.pack(">IH", 0x10000, 0x100) + b"\0" * 10 + mat + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _fullbox(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, n) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) + mat +
                    struct.pack(">II", width << 16, height << 16))
    mdhd = _fullbox(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, 30, n, 0x55C4, 0))
    hdlr = _fullbox(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"pict") + b"\0" * 12 + b"\0")
    vmhd = _fullbox(b"vmhd", 0, 1, b"\0" * 8)
    dinf = _box(b"dinf", _fullbox(b"dref", 0, 0, struct.pack(">I", 1) + _fullbox(b"url ", 0, 1, b"")))
    av1c = _box(b"av1C", av1c_bytes(depth, False, 0, seq_level_idx))
    ccst = _fullbox(b"ccst", 0, 0, struct.pack(">I", 0))
    av01 = _box(b"av01", b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HHIIIH", width, height, 0x480000, 0x480000, 0, 1) +
                b"\0" * 32 + struct.pack(">Hh", 0x18, -1) + av1c + ccst)
    stsd = _fullbox(b"stsd", 0, 0, struct.pack(">I", 1) + av01)
    stts = _fullbox(b"stts", 0, 0, struct.pack(">III", 1, n, 1))
    stsc = _fullbox(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1))
    stsz = _fullbox(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(x)) for x in samples))
    stss = _fullbox(b"stss", 0, 0, struct.pack(">I", len(sync)) + b"".join(struct.pack(">I", k) for k in sync))

    def moov_with(offset):
        stco = _fullbox(b"stco", 0, 0, struct.pack(">II", 1, offset))
        stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco + stss)
        minf = _box(b"minf", vmhd + dinf + stbl)
        mdia = _box(b"mdia", mdhd + hdlr + minf)
        return _box(b"moov", mvhd + _box(b"trak", tkhd + mdia))

    moov = moov_with(0)
    moov = moov_with(len(ftyp) + len(moov) + 8)
    return ftyp + moov + _box(b"mdat", b"".join(bytes(x) for x in samples))


def extract_obus(avif_bytes):
    """Return the payload of the first mdat box (single-item stills only)."""
    i = 0
    while i < len(avif_bytes):
        size, kind = struct.unpack(">I4s", avif_bytes[i:i + 8])
        if kind == b"mdat":
            return avif_bytes[i + 8:i + size]
        i += size
    raise ValueError("no mdat")


# ------------------------------------------------------------------ dav1d decode
def decode_yuv(avif_bytes):
    """Decode with dav1d via libavif; returns (planes, depth) where planes is a list
    of 1 (mono) or 3 numpy arrays (uint8 or uint16) - the exact decoder output."""
    lib = _load()
    dec = lib.avifDecoderCreate()
    img = lib.avifImageCreateEmpty()
    try:
        buf = bytes(avif_bytes)
        r = lib.avifDecoderReadMemory(dec, img, buf, len(buf))
        if r != 0:
            raise RuntimeError("libavif/dav1d decode failed: %s" % lib.avifResultToString(r).decode())
        h = _AvifImageHead.from_address(img)
        depth = h.depth
        planes = []
        fmt = h.yuvFormat  # 1=444 2=422 3=420 4=400
        for p in range(3):
            if not h.yuvPlanes[p]:
                break
            pw = h.width if (p == 0 or fmt == 1) else (h.width + 1) // 2
            ph = h.height if (p == 0 or fmt in (1, 2)) else (h.height + 1) // 2
            rb = h.yuvRowBytes[p]
            raw = C.string_at(h.yuvPlanes[p], rb * ph)
            a = np.frombuffer(raw, dtype=np.uint8).reshape(ph, rb)
            if depth > 8:
                a = a.view(np.uint16)[:, :pw].copy()
            else:
                a = a[:, :pw].copy()
            planes.append(a)
        return planes, depth
    finally:
        lib.avifImageDestroy(img)
        lib.avifDecoderDestroy(dec)


def _readable_ranges():
    out = []
    for line in open("/proc/self/maps"):
        a, perm = line.split()[:2]
        if perm.startswith("r"):
            lo, hi = (int(x, 16) for x in a.split("-"))
            out.append((lo, hi))
    return out


def _image_of_decoder(dec, width, height):
    """`avifDecoder.image`: found by probing the struct's leading pointer-sized slots for an avifImage of the
    expected size (the struct layout differs between libavif versions; nothing is dereferenced blindly)."""
    ranges = _readable_ranges()
    slots = (C.c_uint64 * 32).from_address(dec)
    for v in slots:
        if v and any(lo <= v and v + C.sizeof(_AvifImageHead) <= hi for lo, hi in ranges):
            h = _AvifImageHead.from_address(v)
            if h.width == width and h.height == height and h.depth in (8, 10, 12):
                return h
    raise RuntimeError("avifDecoder.image not found")


def _planes_of(h):
    w, hh, depth = h.width, h.height, h.depth
    out = []
    for p in range(3):
        pw, ph = (w, hh) if p == 0 else ((w + 1) // 2, (hh + 1) // 2)
        rb = h.yuvRowBytes[p]
        raw = C.string_at(h.yuvPlanes[p], rb * ph)
        a = np.frombuffer(raw, dtype=np.uint8).reshape(ph, rb)
        a = a[:, :pw].copy() if depth == 8 else a.view("<u2")[:, :pw].copy()
        out.append(a)
    return out


def decode_sequence(avis_bytes, width, height):
    """Decode every frame of an AVIF image sequence with dav1d (via libavif); returns [planes per frame]."""
    lib = _load()
    lib.avifDecoderSetIOMemory.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.avifDecoderParse.argtypes = [C.c_void_p]
    lib.avifDecoderNextImage.argtypes = [C.c_void_p]
    dec = lib.avifDecoderCreate()
    frames = []
    try:
        buf = bytes(avis_bytes)
        r = lib.avifDecoderSetIOMemory(dec, buf, len(buf))
        if r == 0:
            r = lib.avifDecoderParse(dec)
        if r != 0:
            raise RuntimeError("libavif parse failed: %s" % lib.avifResultToString(r).decode())
        while True:
            r = lib.avifDecoderNextImage(dec)
            if r != 0:
                msg = lib.avifResultToString(r).decode()
                if "No images remaining" in msg or "no images" in msg.lower():
                    break
                raise RuntimeError("libavif/dav1d decode failed at frame %d: %s" % (len(frames), msg))
            frames.append(_planes_of(_image_of_decoder(dec, width, height)))
    finally:
        lib.avifDecoderDestroy(dec)
    return frames


def decode_obus(obus, width, height, depth=8, mono=False):
    return decode_yuv(wrap_avif(obus, width, height, depth, mono))[0]


# ------------------------------------------------------------------ libaom baseline
def libaom_encode_gray(y, cq=30, speed=8, threads=1, extra=None):
    """Encode one 8-bit luma plane as a monochrome AVIF with libaom (end-usage=q)."""
    import io
    from PIL import Image
    adv = {"end-usage": "q", "cq-level": str(cq)}
    if extra:
        adv.update(extra)
    bio = io.BytesIO()
    Image.fromarray(y, "L").save(bio, format="AVIF", codec="aom", speed=speed, max_threads=threads,
                                 subsampling="4:0:0", advanced=adv)
    return bio.getvalue()


def libaom_encode_rgb420(rgb, cq=30, speed=8, threads=1, extra=None):
    import io
    from PIL import Image
    adv = {"end-usage": "q", "cq-level": str(cq)}
    if extra:
        adv.update(extra)
    bio = io.BytesIO()
    Image.fromarray(rgb, "RGB").save(bio, format="AVIF", codec="aom", speed=speed, max_threads=threads,
                                     subsampling="4:2:0", advanced=adv)
    return bio.getvalue()


class _AvifRWData(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_size_t)]


class _AvifEncoderHead(C.Structure):
    # leading (public, settable) fields of `struct avifEncoder` (libavif 1.x avif.h); the defaults read back from
    # avifEncoderCreate() are checked against the documented ones before anything is written (libaom_encode_yuv420)
    _fields_ = [("codecChoice", C.c_int), ("maxThreads", C.c_int), ("speed", C.c_int), ("keyframeInterval", C.c_int), ("timescale", C.c_uint64),
                ("repetitionCount", C.c_int), ("extraLayerCount", C.c_uint32), ("quality", C.c_int), ("qualityAlpha", C.c_int),
                ("minQuantizer", C.c_int), ("maxQuantizer", C.c_int), ("minQuantizerAlpha", C.c_int), ("maxQuantizerAlpha", C.c_int),
                ("tileRowsLog2", C.c_int), ("tileColsLog2", C.c_int), ("autoTiling", C.c_int)]


def libaom_encode_yuv420(frames, depth=8, cq=30, speed=8, threads=1, keyint=1, extra=None):
    """Encode I420 frames ([Y, U, V] numpy planes each, 8 or 10 bit) with libaom through libavif's C API: `end-usage=q`,
    `cq-level=cq` (-> base_q_idx 120 at 30), no RGB conversion anywhere.  One frame -> an AVIF still; several ->
    an AVIF image sequence, key frame every `keyint` frames (1 = all key frames, 0 = only the first: libaom decides).
    Returns the AVIF bytes; decode_yuv / decode_sequence give the exact planes back."""
    lib = _load()
    lib.avifEncoderCreate.restype = C.c_void_p
    lib.avifEncoderDestroy.argtypes = [C.c_void_p]
    lib.avifImageCreate.restype = C.c_void_p
    lib.avifImageCreate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    lib.avifImageAllocatePlanes.argtypes = [C.c_void_p, C.c_uint32]
    lib.avifEncoderAddImage.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    lib.avifEncoderFinish.argtypes = [C.c_void_p, C.POINTER(_AvifRWData)]
    lib.avifEncoderSetCodecSpecificOption.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.avifRWDataFree.argtypes = [C.POINTER(_AvifRWData)]
    lib.avifCodecChoiceFromName.argtypes = [C.c_char_p]
    enc = lib.avifEncoderCreate()
    if not enc:
        raise RuntimeError("avifEncoderCreate failed")
    out = _AvifRWData(None, 0)
    imgs = []
    try:
        h = _AvifEncoderHead.from_address(enc)
        if (h.codecChoice, h.maxThreads, h.speed, h.timescale, h.minQuantizer, h.maxQuantizer) != (0, 1, -1, 1, 0, 63):
            raise RuntimeError("unexpected avifEncoder layout (libavif %s)" % lib.avifVersion().decode())
        h.codecChoice = lib.avifCodecChoiceFromName(b"aom")
        h.maxThreads, h.speed, h.timescale = int(threads), int(speed), 30
        h.keyframeInterval = int(keyint)
        h.minQuantizer, h.maxQuantizer = 0, 63
        adv = {"end-usage": "q", "cq-level": str(cq)}
        if extra:
            adv.update(extra)
        for k, v in adv.items():
            r = lib.avifEncoderSetCodecSpecificOption(enc, k.encode(), str(v).encode())
            if r != 0:
                raise RuntimeError("avifEncoderSetCodecSpecificOption(%s): %s" % (k, lib.avifResultToString(r).decode()))
        single = len(frames) == 1
        for t, planes in enumerate(frames):
            hh, ww = planes[0].shape
            img = lib.avifImageCreate(ww, hh, depth, 3)   # AVIF_PIXEL_FORMAT_YUV420
            imgs.append(img)
            if lib.avifImageAllocatePlanes(img, 1) != 0:   # AVIF_PLANES_YUV
                raise RuntimeError("avifImageAllocatePlanes failed")
            ih = _AvifImageHead.from_address(img)
            ih.yuvRange = 0   # limited, as the Y4M input of the GPU path
            for p in range(3):
                a = np.ascontiguousarray(planes[p].astype(np.uint8 if depth == 8 else "<u2"))
                ph, rb = a.shape[0], ih.yuvRowBytes[p]
                dst = np.frombuffer((C.c_uint8 * (rb * ph)).from_address(ih.yuvPlanes[p]), dtype=np.uint8).reshape(ph, rb)
                dst[:, :a.shape[1] * a.itemsize] = a.view(np.uint8).reshape(ph, -1)
            flags = 2 if single else (1 if (keyint == 1 or t == 0) else 0)   # SINGLE | FORCE_KEYFRAME
            r = lib.avifEncoderAddImage(enc, img, 1, flags)
            if r != 0:
                raise RuntimeError("avifEncoderAddImage: %s" % lib.avifResultToString(r).decode())
        r = lib.avifEncoderFinish(enc, C.byref(out))
        if r != 0:
            raise RuntimeError("avifEncoderFinish: %s" % lib.avifResultToString(r).decode())
        return C.string_at(out.data, out.size)
    finally:
        if out.data:
            lib.avifRWDataFree(C.byref(out))
        for img in imgs:
            lib.avifImageDestroy(img)
        lib.avifEncoderDestroy(enc)


if __name__ == "__main__":
    lib = _load()
    print("libavif", lib.avifVersion().decode())
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:128, 0:192]
    y = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
    av = libaom_encode_gray(y)
    p, d = decode_yuv(av)
    print("libaom mono cq30:", len(av), "bytes; planes", [q.shape for q in p], "depth", d)
    ob = extract_obus(av)
    p2 = decode_obus(ob, 192, 128, 8, mono=True)
    print("re-wrapped identical:", all((a == b).all() for a, b in zip(p, p2)))
