"""mk_index_import_columns_huffman through the C ABI, on blocks made here in numpy: the device inflates deflate blocks of
literals, one lane per block, into rows of the matrix (huff.hip).  The CLI tests cover the path end to end (the binary's
writer produces the blocks, its loader lists them); this file holds the entry point to its contract on hand-made input:
several codes, blocks at odd bit offsets, ragged last blocks, padding slots, the CRC remainders -- and descriptors that
point outside the payload or the rows, name no code, carry an incomplete code or end early: reported, never followed."""
import ctypes as C
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Block(C.Structure):
    _fields_ = [("bit", C.c_uint64), ("out", C.c_uint64), ("out_len", C.c_uint32), ("code", C.c_uint32)]


def rev(code, n):
    return int(format(code, f"0{n}b")[::-1], 2)


def canonical(lens):
    """code (bit-reversed, as the stream holds it) and length per symbol, RFC 1951 3.2.2"""
    count = [0] * 16
    for l in lens:
        count[l] += 1
    count[0] = 0
    nxt, code = [0] * 16, 0
    for l in range(1, 16):
        code = (code + count[l - 1]) << 1
        nxt[l] = code
    out = []
    for l in lens:
        if l:
            out.append((rev(nxt[l], l), l))
            nxt[l] += 1
        else:
            out.append((0, 0))
    return out


def raw_crc(data):
    t = 0
    for b in data:
        t ^= b
        for _ in range(8):
            t = (t >> 1) ^ 0xEDB88320 if t & 1 else t >> 1
    return t


def complete_code_small():
    """16 literals and the end-of-block code: fifteen of 4 bits, the last literal and end-of-block 5 -- exactly complete"""
    lens = [0] * 257
    for v in range(15):
        lens[v] = 4
    lens[15] = 5
    lens[256] = 5
    assert sum(2.0 ** -l for l in lens if l) == 1.0
    return lens


def encode(data, table, bitpos, buf):
    """append the symbols of `data` + end-of-block at bit position bitpos of the bytearray (LSB first)"""
    for s in list(data) + [256]:
        code, n = table[s]
        assert n, s
        for i in range(n):
            if (code >> i) & 1:
                while len(buf) <= (bitpos >> 3):
                    buf.append(0)
                buf[bitpos >> 3] |= 1 << (bitpos & 7)
            bitpos += 1
    while len(buf) <= (bitpos >> 3):
        buf.append(0)
    return bitpos


def run(hip, G, rows, payload, blocks, lens_all, n_codes):
    from miekki_amd import lib as L
    lib = L.load_library()
    ix = hip.Miekki(31, 10, 8, 33, 10)
    try:
        L.check(lib.mk_index_import_begin(ix._h, G))
        arr = (Block * len(blocks))(*blocks)
        crc = (C.c_uint32 * len(blocks))()
        bad = C.c_uint32(12345)
        pl = (C.c_uint8 * len(payload)).from_buffer_copy(bytes(payload))
        ln = (C.c_uint8 * len(lens_all)).from_buffer_copy(bytes(lens_all))
        st = lib.mk_index_import_columns_huffman(ix._h, 0, rows, pl, len(payload), arr, len(blocks), ln, n_codes, crc, C.byref(bad))
        cols = None
        if st == 0:
            cols = np.zeros(rows * G, np.uint8)
            L.check(lib.mk_index_export_columns(ix._h, 0, rows, cols.ctypes.data))
        return st, bad.value, list(crc), cols
    finally:
        ix.close()


def test_blocks_made_by_hand_inflate_into_rows():
    import miekki_amd as hip
    rng = np.random.default_rng(5)
    G, rows = 1000, 37                                               # 37,000 bytes of rows: three 16 KiB-style blocks of our own cutting
    lens_a = [8] * 255 + [9, 9]
    lens_b = complete_code_small()
    ta, tb = canonical(lens_a), canonical(lens_b)
    data = np.concatenate([rng.integers(0, 256, 20_011, dtype=np.uint8), rng.integers(0, 16, rows * G - 20_011, dtype=np.uint8)])
    cuts = [(0, 7_000, 0), (7_000, 13_011, 0), (20_011, 9_989, 1), (30_000, 6_999, 1), (36_999, 1, 1)]
    payload = bytearray()
    bit = 3                                                          # (the first block starts mid-byte, as after a deflate header)
    blocks = []
    group = {0: [], 1: []}
    for at, n, code in cuts:
        start = bit
        bit = encode(data[at:at + n], ta if code == 0 else tb, bit, payload) + 5    # (a gap of header-like bits between blocks)
        group[code].append(Block(start, at, n, code))
    for code in (0, 1):                                              # every 64 slots share a code; empty slots fill a group up
        blocks += group[code] + [Block(0, 0, 0, code)] * (64 - len(group[code]))
    st, bad, crc, cols = run(hip, G, rows, payload, blocks, lens_a + lens_b, 2)
    assert st == 0 and bad == 0
    np.testing.assert_array_equal(cols, data)
    for code in (0, 1):
        for i, b in enumerate(group[code]):
            assert crc[code * 64 + i] == raw_crc(data[b.out:b.out + b.out_len].tobytes())
    # and the remainders fold into zlib's CRC-32 of the whole (what the loader does with fastz's crc32_shift, here by brute force)
    assert zlib.crc32(data.tobytes()) == zlib.crc32(b"".join(data[b.out:b.out + b.out_len].tobytes() for c in (0, 1) for b in group[c]))


@pytest.mark.parametrize("what", ["out past the rows", "out wraps", "bit past the payload", "no such code", "incomplete code", "code too long",
                                  "block ends early", "block runs on"])
def test_bad_descriptors_are_reported_not_followed(what):
    import miekki_amd as hip
    G, rows = 512, 4
    lens = complete_code_small()
    table = canonical(lens)
    data = (np.arange(rows * G) % 16).astype(np.uint8)
    payload = bytearray()
    encode(data, table, 0, payload)
    good = Block(0, 0, rows * G, 0)
    lens_all, n_codes = list(lens), 1
    b = Block(good.bit, good.out, good.out_len, good.code)
    if what == "out past the rows":
        b.out = rows * G - 10
    elif what == "out wraps":
        b.out = (1 << 64) - 8
    elif what == "bit past the payload":
        b.bit = len(payload) * 8 + 64
    elif what == "no such code":
        b.code = 3
    elif what == "incomplete code":
        lens_all[15] = 0
    elif what == "code too long":
        lens_all[15] = 13
    elif what == "block ends early":
        payload = bytearray()
        encode(data[:-3], table, 0, payload)                         # three symbols short: the end-of-block code comes inside the block
    elif what == "block runs on":
        b.out_len = rows * G - 7                                     # no end-of-block where the length says
    blocks = [b] + [Block(0, 0, 0, b.code)] * 63
    st, bad, _, _ = run(hip, G, rows, payload, blocks, lens_all, n_codes)
    assert st == 0 and bad >= 1, (what, st, bad)
