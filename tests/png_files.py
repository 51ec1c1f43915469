"""A minimal PNG writer (8-bit grey, filter 0 or per-row Paeth / Sub / Up filters, one zlib stream) for the host-format tests."""
import struct
import zlib

import numpy as np


def write_png_gray(path, img, filters=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    rows = []
    prev = np.zeros(w, np.int32)
    for y in range(h):
        cur = img[y].astype(np.int32)
        f = (y % 5) if filters else 0
        left = np.concatenate([[0], cur[:-1]]); ul = np.concatenate([[0], prev[:-1]])
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - left
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        rows.append(bytes([f]) + (enc & 0xff).astype(np.uint8).tobytes())
        prev = cur

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(b"".join(rows), 6)) + chunk(b"IEND", b""))
