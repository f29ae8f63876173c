"""Minimal PNG reader / writer on zlib (8-bit gray / RGB / RGBA, non-interlaced) and the numpy restatements of the host logic
that tests/test_host_programs.py and tests/test_gpu_infer_main.py check the C++ host programs against.  Test infrastructure."""
import struct
import zlib

import numpy as np
from scipy import ndimage


def write_png(path, a, filter_type=0):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    color = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    rows = a.reshape(h, w * c).astype(np.int16)
    if filter_type == 0:
        filt = rows
    elif filter_type == 1:      # Sub
        filt = rows.copy(); filt[:, c:] -= rows[:, :-c]
    elif filter_type == 2:      # Up
        filt = rows.copy(); filt[1:] -= rows[:-1]
    else:
        raise ValueError(filter_type)
    raw = b"".join(bytes([filter_type]) + (r & 255).astype(np.uint8).tobytes() for r in filt)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(b):
        n, t = struct.unpack(">I4s", b[pos:pos + 8])
        body = b[pos + 8:pos + 8 + n]
        if t == b"IHDR":
            w, h, depth, color, _, _, inter = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and inter == 0
        elif t == b"IDAT":
            idat += body
        pos += 12 + n
    c = {0: 1, 2: 3, 4: 2, 6: 4}[color]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * c + 1)
    out = np.zeros((h, w * c), np.uint8)
    for y in range(h):
        f, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(w * c, np.int32)
        cur = np.zeros(w * c, np.int32)
        if f in (0, 2):
            cur = (line + (up if f == 2 else 0)) & 255
        else:
            for i in range(w * c):
                a_ = cur[i - c] if i >= c else 0
                b_ = up[i]
                c_ = up[i - c] if i >= c else 0
                if f == 1:
                    p = a_
                elif f == 3:
                    p = (a_ + b_) >> 1
                else:
                    pa, pb, pc = abs(b_ - c_), abs(a_ - c_), abs(a_ + b_ - 2 * c_)
                    p = a_ if pa <= pb and pa <= pc else (b_ if pb <= pc else c_)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
    return out.reshape(h, w, c) if c > 1 else out.reshape(h, w)


DEFAULT_CLASSES = [(0, 255, 0, 64), (255, 255, 0, 128), (255, 0, 0, 128)]   # annonet_parse_anno_classes.cpp:25-29


def labels_to_rgba(labels, classes=DEFAULT_CLASSES):
    out = np.zeros(labels.shape + (4,), np.uint8)    # (0,0,0,0) = ignore
    for k, col in enumerate(classes):
        out[labels == k] = col
    return out


def resize_nearest(a, tw, th):
    """dlib resize_image + interpolate_nearest_neighbor as annonet_host.h restates it: corner-aligned grid, floor(v + 0.5)"""
    nr, nc = a.shape[:2]
    ys = np.floor(np.arange(th) * ((nr - 1) / max(th - 1, 1)) + 0.5).astype(int)
    xs = np.floor(np.arange(tw) * ((nc - 1) / max(tw - 1, 1)) + 0.5).astype(int)
    return a[ys][:, xs]


def blobs_8(img):
    """label_connected_blobs(zero background, 8 neighbours, connected_if_equal): 0 = background, else a distinct id per region"""
    out = np.zeros(img.shape, np.int64)
    nxt = 1
    for v in np.unique(img):
        if v == 0:
            continue
        lab, n = ndimage.label(img == v, structure=np.ones((3, 3)))
        out[lab > 0] = lab[lab > 0] + nxt - 1
        nxt += n
    return out, nxt


def confusion_matrices(gt, res, K):
    """annonet_infer_main.cpp:482-492 (per pixel) and :202-272 (per region, two-way); ties -> the smallest class index"""
    per_pixel = np.zeros((K, K), np.int64)
    valid = gt != 65535
    np.add.at(per_pixel, (gt[valid], res[valid]), 1)
    per_region = np.zeros((K, K), np.int64)
    if not valid.any():
        return per_pixel, per_region

    def winner(votes):
        return 65535 if not votes else min(votes, key=lambda k: (-votes[k], k))
    for blobs, n in (blobs_8(gt), blobs_8(res)):
        for b in range(n):
            m = (blobs == b) & valid
            if not m.any():
                continue
            g, gc = np.unique(gt[m], return_counts=True)
            p, pc = np.unique(res[m], return_counts=True)
            vg, vp = dict(zip(g.tolist(), gc.tolist())), dict(zip(p.tolist(), pc.tolist()))
            if winner(vg) != 0 and not (len(vp) == 1 and 0 in vp):
                vp.pop(0, None)
            per_region[winner(vg), winner(vp)] += 1
    return per_pixel, per_region
