"""The host-side packer of the packed ingest format (include/miekki_hip.h: mk_packed_seq,
mk_pack_append) against a plain numpy statement of the format.  No GPU involved: the packer is pure
host code inside libmiekki_hip.so."""
import numpy as np
import pytest

import miekki_amd
from miekki_amd.index import pack_sequence


def reference_pack(seq: bytes):
    """codes: base i at bits 2*(i%32) of word i//32, A C G T -> 0 1 2 3, anything else 0 (nuc2int,
    utils.cpp:31-49); except: bit i%64 of word i//64 where the character is not one of ACGT."""
    a = np.frombuffer(seq, np.uint8)
    code = np.zeros(a.size, np.uint64)
    for ch, v in ((b"C", 1), (b"G", 2), (b"T", 3)):
        code[a == ch[0]] = v
    bad = ~np.isin(a, np.frombuffer(b"ACGT", np.uint8))
    codes = np.zeros((a.size + 31) // 32, np.uint64)
    exc = np.zeros((a.size + 63) // 64, np.uint64)
    idx = np.arange(a.size)
    np.bitwise_or.at(codes, idx // 32, code << ((idx % 32) * 2).astype(np.uint64))
    np.bitwise_or.at(exc, idx // 64, bad.astype(np.uint64) << (idx % 64).astype(np.uint64))
    return codes, exc, bool(bad.any())


def random_sequence(rng, n, junk):
    alphabet = np.frombuffer(b"ACGT" * 8 + (b"NnacgtRY-*\x00\xff" if junk else b""), np.uint8)
    return bytes(rng.choice(alphabet, n))


@pytest.mark.parametrize("piece", [None, 1, 7, 31, 32, 33, 60, 80, 1000])
def test_pack_append_matches_the_format(piece):
    rng = np.random.default_rng(3 if piece is None else piece)
    for n in (0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 1000, 4097, 10_001):
        for junk in (False, True):
            seq = random_sequence(rng, n, junk)
            codes, exc, m, head = pack_sequence(seq, piece)
            want_c, want_x, dirty = reference_pack(seq)
            assert m == n and head == seq[:32]
            np.testing.assert_array_equal(codes[:want_c.size], want_c)
            assert not codes[want_c.size:].any()                       # the slack words stay zero
            if dirty:
                np.testing.assert_array_equal(exc[:want_x.size], want_x)
                assert not exc[want_x.size:].any()
            else:
                assert exc is None


def test_pack_append_keeps_what_lies_below_and_clears_what_lies_above():
    lib = miekki_amd.load_library()
    codes = np.full(8, 0xFFFFFFFFFFFFFFFF, np.uint64)                   # garbage everywhere: nothing needs clearing first
    exc = np.full(8, 0xFFFFFFFFFFFFFFFF, np.uint64)
    first, second = b"ACGTN" * 9, b"TTGCAacgt" * 11                      # 45 + 99 characters
    assert lib.mk_pack_append(codes.ctypes.data, exc.ctypes.data, 0, first, len(first)) == 1
    assert lib.mk_pack_append(codes.ctypes.data, exc.ctypes.data, len(first), second, len(second)) == 1
    want_c, want_x, _ = reference_pack(first + second)
    np.testing.assert_array_equal(codes[:want_c.size], want_c)
    np.testing.assert_array_equal(exc[:want_x.size], want_x)
    assert lib.mk_pack_append(None, exc.ctypes.data, 0, first, 1) == -1
    assert lib.mk_pack_code_words(64) >= 2 and lib.mk_pack_except_words(64) >= 1


def test_scalar_packer_equals_the_vector_one():
    """MIEKKI_PACK_SCALAR=1 (read once per process) selects the portable loop: run it in a child."""
    import os, subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, test_host_pack as t\n"
            "rng = np.random.default_rng(11)\n"
            "for n in (1, 33, 64, 65, 5000):\n"
            "    s = t.random_sequence(rng, n, True)\n"
            "    c, x, m, h = t.pack_sequence(s, 61)\n"
            "    wc, wx, d = t.reference_pack(s)\n"
            "    assert (c[:wc.size] == wc).all() and (x is None) == (not d) and (x is None or (x[:wx.size] == wx).all())\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MIEKKI_PACK_SCALAR="1"), capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
