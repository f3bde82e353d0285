"""gzip'd input inflated on the GPU (the reference reads genome files through zstr / zlib,
Miekki.cpp:559-567).  mk_gz_inflate against zlib on streams of every block type, level and shape -- and hostile ones
(truncated, a distance beyond the output, over-subscribed and incomplete codes, bad stored lengths, trailing bytes, a
wrong CRC or size), which must come back with a status, never with text and never with a hang."""
import gzip
import struct
import zlib

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

OK, EMPTY, NOT_GZIP, TRUNCATED, BAD_BLOCK, BAD_STORED, BAD_LENGTHS, BAD_CODE, BAD_DISTANCE, TOKEN_ROOM, OUTPUT_ROOM, TRAILING, BAD_CRC, BAD_SIZE = range(14)


@pytest.fixture(scope="module")
def ctx():
    import miekki_amd
    ix = miekki_amd.Miekki(21, 10, 8, 32, 10)
    yield ix
    ix.close()


def inflate(ctx, blobs, rooms=None):
    from miekki_amd import lib as L
    if rooms is None:
        rooms = [len(gzip.decompress(b)) + 64 for b in blobs]
    return L.gz_inflate(ctx._h, blobs, rooms)


def ok_by_zlib(b):
    try:
        d = zlib.decompressobj(47)
        d.decompress(b)
        return d.eof
    except zlib.error:
        return False


def gz(data, level=6, **kw):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, **kw)
    return c.compress(data) + c.flush()


def fasta(n, seed=0, width=80):
    return synth.fasta(f"genome{seed}", synth.genome_bases(1000 + seed, 0, n), width) if width != 80 else synth.fasta(f"genome{seed}", synth.genome_bases(1000 + seed, 0, n))


def test_streams_of_every_kind_against_zlib(ctx):
    rng = np.random.default_rng(3)
    texts = [
        b"", b"A", b"ACGT" * 5, fasta(100), fasta(70_000), fasta(300_001, 1),
        bytes(rng.integers(0, 256, 50_000, dtype=np.uint8)),                          # incompressible: stored blocks at some levels
        b"N" * 200_000,                                                               # one long run: distance 1, length 258
        (b">h\n" + b"ACGTTGCAAC" * 7 + b"\n") * 3000,                                  # periodic
        b"".join(bytes([65 + int(x)]) for x in rng.integers(0, 26, 40_000)),          # text-like: many literals, long codes
        bytes(rng.integers(0, 4, 123_457, dtype=np.uint8) + 65),
    ]
    blobs, want = [], []
    for t in texts:
        for level in (0, 1, 6, 9):
            blobs.append(gz(t, level)); want.append(t)
        blobs.append(gz(t, 6, strategy=zlib.Z_FIXED)); want.append(t)                 # fixed Huffman blocks
        blobs.append(gz(t, 6, strategy=zlib.Z_HUFFMAN_ONLY)); want.append(t)          # literals only, no distance code at all
        blobs.append(gz(t, 6, strategy=zlib.Z_RLE)); want.append(t)                   # distance 1 only: a one-code distance alphabet
        blobs.append(gz(t, 9, memLevel=1)); want.append(t)                            # many small blocks
    got, status = inflate(ctx, blobs)
    for i, (g, s, w) in enumerate(zip(got, status, want)):
        assert s == OK, (i, s, len(w))
        assert g == w, (i, len(g), len(w))


def test_members_headers_and_sync_points(ctx):
    a, b, c = fasta(50_000, 2), fasta(10, 3), b""
    multi = gz(a) + gz(b, 1) + gz(c) + gz(a[:777], 9)                                 # members back to back (bgzip, pigz -i, cat)
    hdr = struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 0, 0, 3, 5) + b"extra" + b"name.fa\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = body.compress(a) + body.flush()
    decorated = hdr + raw + struct.pack("<II", zlib.crc32(a), len(a))                  # every optional header field
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    flushed = co.compress(a[:20_000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(a[20_000:]) + co.flush(zlib.Z_FULL_FLUSH) + co.flush()
    got, status = inflate(ctx, [multi, decorated, flushed, gzip.compress(a, 6)])
    assert status == [OK] * 4
    assert got[0] == a + b + c + a[:777] and got[1] == a and got[2] == a and got[3] == a


def deflate_bits(bits):
    """a raw deflate stream from a list of (value, nbits), LSB first"""
    acc = n = 0
    out = bytearray()
    for v, k in bits:
        acc |= v << n; n += k
        while n >= 8:
            out.append(acc & 255); acc >>= 8; n -= 8
    if n:
        out.append(acc & 255)
    return bytes(out)


def wrap(raw, text=b""):
    return b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + raw + struct.pack("<II", zlib.crc32(text), len(text))


def test_hostile_streams_are_reported_not_followed(ctx):
    a = fasta(60_000, 5)
    good = gz(a)
    cases = {}
    cases["truncated in the data"] = (good[:len(good) // 2], {TRUNCATED, BAD_CODE, BAD_LENGTHS})
    cases["truncated in the trailer"] = (good[:-3], {TRUNCATED})
    cases["no trailer"] = (good[:-8], {TRUNCATED})
    cases["too short"] = (good[:10], {NOT_GZIP})
    cases["not gzip"] = (b"ACGT" * 100, {NOT_GZIP})
    cases["zlib wrapper"] = (zlib.compress(a), {NOT_GZIP})
    cases["reserved flag"] = (good[:3] + b"\x20" + good[4:], {NOT_GZIP})
    cases["trailing garbage"] = (good + b"garbage that is no member" * 2, {TRAILING})
    cases["trailing zeros"] = (good + b"\0" * 40, {TRAILING})
    cases["wrong crc"] = (good[:-8] + struct.pack("<I", zlib.crc32(a) ^ 1) + good[-4:], {BAD_CRC})
    cases["wrong size"] = (good[:-4] + struct.pack("<I", len(a) + 1), {BAD_SIZE})
    flipped = bytearray(good); flipped[len(good) // 3] ^= 0x10
    cases["a flipped bit"] = (bytes(flipped), set(range(1, 14)) - {EMPTY})
    cases["block type 3"] = (wrap(deflate_bits([(1, 1), (3, 2)])), {BAD_BLOCK})
    cases["stored length check"] = (wrap(deflate_bits([(1, 1), (0, 2), (0, 5), (5, 16), (5, 16)]) + b"hello"), {BAD_STORED})
    # fixed block: a match at distance 1 with nothing before it (length code 257 = 0000001, distance code 0 = 00000)
    cases["distance beyond the start"] = (wrap(deflate_bits([(1, 1), (1, 2), (0b1000000, 7), (0, 5), (0, 7)])), {BAD_DISTANCE})
    # fixed block: literal/length symbols 286 and 287 exist in the code (11000110, 11000111 reversed) but are invalid
    cases["symbol 286"] = (wrap(deflate_bits([(1, 1), (1, 2), (0b01100011, 8)])), {BAD_CODE})
    # fixed block: distance codes 30 / 31
    cases["distance code 30"] = (wrap(deflate_bits([(1, 1), (1, 2), (0b0001100, 8), (0b1000000, 7), (0b01111, 5)]), b""), {BAD_CODE, BAD_DISTANCE})
    # dynamic block whose code length code is over-subscribed: all nineteen lengths 1
    cases["over-subscribed code length code"] = (wrap(deflate_bits([(1, 1), (2, 2), (0, 5), (0, 5), (15, 4)] + [(1, 3)] * 19)), {BAD_LENGTHS})
    # ... incomplete: one code of length 2 only
    cases["incomplete code length code"] = (wrap(deflate_bits([(1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (2, 3), (0, 3), (0, 3), (0, 3)])), {BAD_LENGTHS})
    # dynamic block: repeat (symbol 16) with nothing before it.  code length code: symbols 16 and 0 with 1 bit each
    cases["repeat without a length before it"] = (wrap(deflate_bits([(1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (1, 3), (0, 3), (0, 3), (1, 3), (1, 1), (0, 2)])), {BAD_LENGTHS})
    # dynamic block: HLIT = 31 -> 288 literal / length codes (more than 286)
    cases["too many length codes"] = (wrap(deflate_bits([(1, 1), (2, 2), (31, 5), (0, 5), (0, 4)] + [(0, 3)] * 4)), {BAD_LENGTHS})
    cases["output beyond the room"] = (gz(b"N" * 100_000), {OUTPUT_ROOM})
    names = list(cases)
    blobs = [cases[k][0] for k in names]
    rooms = [1 << 16 if k != "output beyond the room" else 50_000 for k in names]
    rooms[names.index("a flipped bit")] = len(a) + 4096
    got, status = inflate(ctx, blobs + [good], rooms + [len(a)])
    for k, g, s in zip(names, got, status):
        assert s in cases[k][1], (k, s)
        assert g is None, k
        if s != OK:
            assert not ok_by_zlib(cases[k][0]) or k in ("output beyond the room", "trailing zeros", "trailing garbage", "zlib wrapper"), k   # nothing zlib takes is refused here but by room, tail or wrapper
    assert status[-1] == OK and got[-1] == a                        # (a good stream among them is not disturbed; room exactly its size)


def test_many_streams_of_uneven_length(ctx):
    """more streams than one wave, lengths from 0 to 400 kB: lanes finish at different times, blocks change out of step"""
    rng = np.random.default_rng(9)
    texts = [fasta(int(n), 10 + i) if i % 3 else bytes(rng.integers(65, 70, int(n), dtype=np.uint8)) for i, n in enumerate(rng.integers(0, 400_000, 150))]
    blobs = [gz(t, int(rng.choice([1, 4, 6, 9]))) for t in texts]
    got, status = inflate(ctx, blobs)
    assert status == [OK] * len(texts)
    assert got == texts


# ---- gzip'd genome FILES -> the index (mk_gz_unpack: inflate + strip on the device, the sequences never visit the host)
import hashlib
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,level", [(n, l) for n in ["messy", "h20", "w16", "rnd0", "rnd2", "rnd3", "rnd4", "rnd5"] for l in (6, 1)
                                        if l == 6 or n in ("messy", "w16", "rnd2", "rnd5")])     # (level 1 for half of them: the suite's time)
def test_index_from_gzipped_files_equals_the_reference(name, level):
    """Every genome file of a reference-pinned case handed over gzip'd (whatever it was on disk): the index stream -- columns,
    sizes, Bloom bytes -- must be the one the real reference built from the same files (tests/golden/*.npz)."""
    import miekki_amd
    case = synth.CASES[name]()
    gold = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ix = miekki_amd.Miekki(case.k, case.h, case.fp_bits, case.b, case.threshold)
    try:
        blobs = []
        for _, text, _ in case.genome_files:
            blobs.append(gz(text, level) if len(text) % 3 else gz(text[:len(text) // 2], level) + gz(text[len(text) // 2:], 9))   # (some as two members)
        status = ix.insert_gz_files(blobs, fallback=False)
        assert status == [OK] * len(blobs), status                   # every file, one member or two, is the device's (no fallback to hide behind)
        assert ix.index_size == int(gold["G"])
        np.testing.assert_array_equal(ix.sketch_size, gold["sketch_size"])
        np.testing.assert_array_equal(ix.genome_size, gold["genome_size"])
        raw = bytearray(b"".join(ix.serialize()))
        raw[32] = 0; raw[38] = 0
        assert hashlib.sha256(bytes(raw)).hexdigest() == str(gold["stream_sha_masked"])
    finally:
        ix.close()


def test_fasta_shapes_and_fallback_against_oracle():
    """Texts that stress the strip: no trailing line feed, empty lines, header-only files, a header as the last line without
    a line feed, '>' inside a line, carriage returns, lines longer than a chunk, a line feed exactly at a chunk's edge, a
    header line across a chunk's edge -- and files the device refuses among them (their sequences come from the host)."""
    import miekki_amd
    from oracle import oracle as orc
    k, h = 15, 10
    rng = np.random.default_rng(17)
    base = [synth.genome_bases(700 + i, 0, 30_000) for i in range(6)]
    texts = [
        b">a\n" + base[0],                                                            # no line feed at the end
        b">a\n\n\n" + base[1][:9000] + b"\n\n>b desc\n" + base[1][9000:] + b"\n\n",   # empty lines, two records
        b">only a header\n",
        b">h\n" + base[2][:5000] + b"\n>last header without line feed",
        base[3][:100] + b">notaheader" + base[3][100:4000] + b"\n" + b"ACGT>ACGT\n",   # '>' inside lines
        b">crlf\r\n" + b"\r\n".join(base[4][i:i + 60] for i in range(0, 12_000, 60)) + b"\r\n",
        b">long line\n" + base[5] + b"\n",                                            # a 30,000-byte line
        b">edge\n" + base[0][:4096 - 6 - 1] + b"\n" + base[0][5000:9000] + b"\n",    # line feed is the chunk's last byte
        b">edge2\n" + base[1][:4096 - 7 - 3] + b"\n" + b">" + b"x" * 50 + b"\n" + base[1][:3000] + b"\n",   # a header across the edge
        b"".join((b">c%d\n" % i) + base[i % 6][i * 10:i * 10 + int(rng.integers(1, 200))] + b"\n" for i in range(300)),   # 300 tiny records
        b"",
        b"\n\n\n",
    ]
    blobs = [gz(t, 6) for t in texts]
    blobs[1] = blobs[1] + b"trailing bytes"                                          # refused (trailing): the host's inflater takes it
    texts[1] = texts[1]
    blobs.append(b"not gzip at all, but a long enough run of bytes"); texts.append(None)
    seqs = [b"".join(l for l in t.split(b"\n") if not l.startswith(b">")) for t in texts if t is not None]
    o = orc.OracleMiekki(k, h, 8, 32, 10)
    o.insert_sequences([s_ for s_ in seqs if len(s_) >= k])
    ix = miekki_amd.Miekki(k, h, 8, 32, 10)
    try:
        status = ix.insert_gz_files(blobs[:-1], fallback=True)
        assert status[1] == TRAILING and all(s_ == OK for i, s_ in enumerate(status) if i != 1), status
        assert ix.insert_gz_files([blobs[-1]], fallback=False) == [NOT_GZIP]
        assert ix.index_size == o.index_size
        np.testing.assert_array_equal(ix.sketch_size, o.sketch_size)
        np.testing.assert_array_equal(ix.genome_size, o.genome_size)
        a = bytearray(b"".join(ix.serialize())); a[32] = 0
        b = bytearray(o.serialize().tobytes()); b[32] = 0
        assert bytes(a) == bytes(b)
    finally:
        ix.close()


def test_a_text_beyond_a_gibibyte_is_the_device_s_too(ctx):
    """zstr has no limit on a file's text (zstr.hpp:186-190); the device's inflater counts in 32 bits: a text of 1.1 GiB in one
    file -- seventy members of 16 MiB, as bgzip or `cat` would leave them -- comes back whole, CRC and length of every
    member checked on the device."""
    # (a 16 KiB block over and over on one line: matches of 258 bytes -- a wave takes 64 tokens a step whatever their length,
    # so this text costs a fortieth of the steps random DNA of the same length would)
    part = b">big\n" + synth.genome_bases(4242, 0, 16 * 1024) * 1024 + b"\n"
    member = gz(part, 1)
    members = 70
    blob = member * members
    assert len(part) * members > (1 << 30)
    got, status = inflate(ctx, [blob], rooms=[len(part) * members])
    assert status == [OK]
    assert len(got[0]) == len(part) * members
    crc = 0
    for k in range(members):
        assert got[0][k * len(part):k * len(part) + 4096] == part[:4096]
        crc = zlib.crc32(got[0][k * len(part):(k + 1) * len(part)])
        assert crc == zlib.crc32(part), k
