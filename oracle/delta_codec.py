"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (nothing upstream to pin against, see below).

numpy restatement of the delta-snapshot stream "NBD1" (nbody-simulation_amd/csrc/delta_codec.h).  Upstream has no
such format: /root/reference src/main.rs:107-134 is a commented-out experiment that subtracts the positions across an
update and prints the zstd size of the raw bytes.  The format is this project's; this file is its independent,
straight-line statement, used by tests/ to check the device encoder byte for byte and the host decoder bit for bit.
Only tests/ may import it.
"""
from __future__ import annotations

import struct

import numpy as np

HEADER = 32


def _types(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return np.uint32, np.int32, 32
    if dtype == np.float64:
        return np.uint64, np.int64, 64
    raise ValueError("f32 or f64")


def keys_of(pos):
    """Ordered-integer keys of the coordinates' bit patterns: (2, npad) array, x row then y row, zero padded."""
    pos = np.ascontiguousarray(pos)
    U, _, bits = _types(pos.dtype)
    n = pos.shape[0]
    npad = (n + 63) // 64 * 64
    u = pos.view(U).reshape(n, 2)
    sign = (u >> U(bits - 1)).astype(bool)
    k = np.where(sign, ~u, u | (U(1) << U(bits - 1)))
    out = np.zeros((2, npad), U)
    out[:, :n] = k.T
    return out


def pos_of(keys, n, dtype):
    U, _, bits = _types(dtype)
    k = keys[:, :n].T.copy()
    top = (k >> U(bits - 1)).astype(bool)
    u = np.where(top, k & ~(U(1) << U(bits - 1)), ~k)
    return np.ascontiguousarray(u.astype(U)).view(dtype).reshape(n, 2)


def _zigzag(r, S, U, bits):
    return ((r << U(1)) ^ (r.view(S) >> S(bits - 1)).view(U)).astype(U)


def _unzigzag(z, U):
    return ((z >> U(1)) ^ (U(0) - (z & U(1)))).astype(U)


def _width(z):
    m = int(z.max()) if z.size else 0
    return m.bit_length()


class Encoder:
    """State = keys of the last two snapshots (zero after reset)."""

    def __init__(self):
        self.prev = self.prev2 = None
        self.key_next = True

    def reset(self):
        self.key_next = True

    def encode(self, pos, step=0) -> bytes:
        pos = np.ascontiguousarray(pos)
        U, S, bits = _types(pos.dtype)
        n = pos.shape[0]
        cur = keys_of(pos)
        npad = cur.shape[1]
        nblk = npad // 64
        if self.prev is None or self.prev.shape != cur.shape or self.prev.dtype != cur.dtype:
            self.key_next = True
        key = self.key_next
        if key:
            self.prev = np.zeros_like(cur)
            self.prev2 = np.zeros_like(cur)
        with np.errstate(over="ignore"):
            z0 = _zigzag(cur - self.prev, S, U, bits)
            z1 = _zigzag(cur - (self.prev + (self.prev - self.prev2)), S, U, bits)
        widths = np.zeros((2 * nblk + 7) // 8 * 8, np.uint8)
        words = []
        for blk in range(nblk):
            for c in range(2):
                a, b = z0[c, blk * 64:blk * 64 + 64], z1[c, blk * 64:blk * 64 + 64]
                w0, w1 = _width(a), _width(b)
                pred = 1 if w1 < w0 else 0
                w, z = (w1, b) if pred else (w0, a)
                widths[2 * blk + c] = w | (pred << 7)
                for bit in range(w):
                    plane = 0
                    for lane in range(64):
                        plane |= ((int(z[lane]) >> bit) & 1) << lane
                    words.append(plane)
        self.prev2, self.prev = self.prev, cur
        self.key_next = False
        head = b"NBD1" + bytes([bits, 1 if key else 0, 0, 0]) + struct.pack("<QQQ", n, step, len(words))
        return head + widths.tobytes() + struct.pack("<%dQ" % len(words), *words)


class Decoder:
    def __init__(self):
        self.prev = self.prev2 = None
        self.n = -1
        self.dtype = None
        self.step = 0

    def apply(self, stream: bytes):
        if len(stream) < HEADER or stream[:4] != b"NBD1":
            raise ValueError("bad stream")
        bits, key = stream[4], stream[5]
        n, step, total = struct.unpack("<QQQ", stream[8:32])
        dtype = np.float32 if bits == 32 else np.float64
        U, _, _ = _types(dtype)
        nblk = (n + 63) // 64
        npad = nblk * 64
        wb = (2 * nblk + 7) // 8 * 8
        if len(stream) != HEADER + wb + 8 * total:
            raise ValueError("size mismatch")
        widths = np.frombuffer(stream, np.uint8, wb, HEADER)
        words = struct.unpack("<%dQ" % total, stream[HEADER + wb:])
        if key:
            self.prev = np.zeros((2, npad), U)
            self.prev2 = np.zeros((2, npad), U)
        elif self.n != n or self.dtype != dtype:
            raise ValueError("out of sequence")
        cur = np.zeros((2, npad), U)
        at = 0
        mask = (1 << bits) - 1
        for blk in range(nblk):
            for c in range(2):
                wbyte = int(widths[2 * blk + c])
                w = wbyte & 127
                planes = words[at:at + w]
                at += w
                for lane in range(64):
                    z = 0
                    for bit in range(w):
                        z |= ((planes[bit] >> lane) & 1) << bit
                    r = (z >> 1) ^ (-(z & 1) & mask)
                    p1, p2 = int(self.prev[c, blk * 64 + lane]), int(self.prev2[c, blk * 64 + lane])
                    pred = (2 * p1 - p2) & mask if wbyte & 128 else p1
                    cur[c, blk * 64 + lane] = (pred + r) & mask
        self.prev2, self.prev = self.prev, cur
        self.n, self.dtype, self.step = n, dtype, step

    def positions(self):
        return pos_of(self.prev, self.n, self.dtype)
