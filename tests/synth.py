"""Deterministic test inputs (SURVEY.md section 8d generator) and the named parity cases.

Used by tests/golden/make_golden.py (to feed the real reference) and by the tests
(to feed the oracle and the HIP path the same bytes).  Fixtures under
tests/golden/ hold only the reference's OUTPUTS for these inputs.
"""
from __future__ import annotations

import numpy as np

SEED_G = 0x4D49454B4B490001
SEED_Q = 0x4D49454B4B490002
SEED_M = 0x4D49454B4B490003
_M64 = (1 << 64) - 1
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(z):
    """Vectorised splitmix64 on uint64 arrays (wrap-around arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def splitmix64_int(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def genome_bases(g: int, off: int, n: int) -> bytes:
    """n bases of synthetic genome g starting at position off."""
    w0, w1 = off >> 5, (off + n + 31) >> 5
    idx = np.arange(w0, w1, dtype=np.uint64)
    words = splitmix64(np.uint64(SEED_G) ^ (np.uint64(g) << np.uint64(32)) ^ idx)
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    codes = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
    lo = off - (w0 << 5)
    return _ACGT[codes[lo:lo + n]].tobytes()


def query_origin(q: int, G: int, L: int, qlen: int):
    return q % G, splitmix64_int(SEED_Q ^ q) % (L - qlen)


def mutate(seq: bytes, q: int, rate: float) -> bytes:
    """Substitute round(rate*len) positions (position/base from SEED_M stream)."""
    s = bytearray(seq)
    for j in range(int(round(rate * len(s)))):
        r = splitmix64_int(SEED_M ^ (q << 20) ^ j)
        pos = r % len(s)
        s[pos] = b"ACGT"[((b"ACGT".index(bytes([s[pos]])) if bytes([s[pos]]) in b"ACGT" else 0)
                          + 1 + ((r >> 40) % 3)) % 4]
    return bytes(s)


def strain(species: int, strain_id: int, length: int, rate: float) -> bytes:
    """Strain `strain_id` of synthetic species `species`: the species' genome (genome_bases(species, 0, length)) with
    round(rate * length) substitutions at seeded positions (strain 0 = the species genome itself).  Strains of one
    species are independent descendants of it: two of them differ in about twice as many places.  A collection of
    related strains is what the reference's README has in mind for its column compression (README.md:136-138)."""
    seq = np.frombuffer(genome_bases(species, 0, length), np.uint8).copy()
    n = int(round(rate * length)) if strain_id else 0
    if n:
        rng = np.random.default_rng((species << 20) ^ (strain_id << 1) ^ 0x5eed)
        pos = rng.integers(0, length, n)
        code = (np.searchsorted(_ACGT, seq[pos]) + rng.integers(1, 4, n)) & 3      # always a different base
        seq[pos] = _ACGT[code]
    return seq.tobytes()


SEED_S = 0x4D49454B4B490004


def strain_device(g: int, strains: int, rate_ppm: int, off: int, n: int) -> bytes:
    """n bases from position off of genome g of the DEVICE's strain generator (mk_index_append_synthetic_strains;
    mk_device.hpp: strain_word): strain g % strains of species g // strains -- the species' genome is the synthetic genome
    `species` itself, every other strain carries per 32-base word, with probability 32 x rate, one substitution."""
    species, t = divmod(g, strains)
    w0, w1 = off >> 5, (off + n + 31) >> 5
    idx = np.arange(w0, w1, dtype=np.uint64)
    words = splitmix64(np.uint64(SEED_G) ^ (np.uint64(species) << np.uint64(32)) ^ idx)
    if t:
        r = splitmix64(np.uint64(SEED_S) ^ (np.uint64(g) << np.uint64(32)) ^ idx)
        hit = (r & np.uint64(0xffffffff)) < np.uint64(rate_ppm * 32 * 4295)
        sh = np.uint64(62) - np.uint64(2) * ((r >> np.uint64(32)) & np.uint64(31))
        old = (words >> sh) & np.uint64(3)
        new = (old + np.uint64(1) + ((r >> np.uint64(40)) % np.uint64(3))) & np.uint64(3)
        words = np.where(hit, (words & ~(np.uint64(3) << sh)) | (new << sh), words)
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    codes = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
    lo = off - (w0 << 5)
    return _ACGT[codes[lo:lo + n]].tobytes()


def tandem_rich(g: int, length: int, share: float, unit_lo: int = 2, unit_hi: int = 60) -> bytes:
    """Genome g with about `share` of its length in tandem repeats and homopolymer runs (seeded): runs of 200-5,000
    bases made of a 1-60 base unit, between stretches of the usual random sequence.  Repeat-rich input is what makes
    many k-mers of one stretch fall into one partition (the build's reduce kernel: many lanes on one LDS entry)."""
    rng = np.random.default_rng(0xBEEF ^ (g << 8))
    out = bytearray(genome_bases(g, 0, length))
    filled = 0
    while filled < share * length:
        n = int(rng.integers(200, 5000))
        at = int(rng.integers(0, length - n))
        unit = 1 if rng.random() < 0.3 else int(rng.integers(unit_lo, unit_hi + 1))
        u = bytes(rng.choice(list(b"ACGT"), unit).tolist())
        out[at:at + n] = (u * (n // unit + 1))[:n]
        filled += n
    return bytes(out)


def fasta(name: str, seq: bytes, width: int = 80) -> bytes:
    lines = [b">" + name.encode()]
    lines += [seq[i:i + width] for i in range(0, len(seq), width)]
    return b"\n".join(lines) + b"\n"


class Case:
    """One named parity case: parameters + genome files + query records."""

    def __init__(self, name, k, h, f, b, threshold):
        self.name, self.k, self.h, self.f, self.b, self.threshold = name, k, h, f, b, threshold
        self.genome_files = []   # (file name, file bytes, gzip?)
        self.queries = []        # (header bytes incl '>', sequence bytes)

    @property
    def fp_bits(self):
        return 5 + self.f

    def genome_sequences(self):
        """Concatenated non-'>' lines per file, as index_file_of_file builds them
        (Miekki.cpp:559-567); files whose sequence is shorter than k are skipped (569)."""
        out = []
        for _, data, _ in self.genome_files:
            seq = b"".join(l for l in data.split(b"\n") if not l.startswith(b">"))
            if len(seq) >= self.k:
                out.append(seq)
        return out

    def query_sequences(self):
        """Records query_file keeps (Miekki.cpp:465): sequence at least k long."""
        return [(h, s) for h, s in self.queries if len(s) >= self.k]


def _std_queries(case, G, L, n, qlen=1000, start=0):
    for q in range(start, start + n):
        g, off = query_origin(q, G, L, qlen)
        case.queries.append((f">q{q}_g{g}_p{off}".encode(), genome_bases(g, off, qlen)))


def case_c1() -> Case:
    """BASELINE config 1: 10 synthetic 5 Mb genomes, -k 31 -h 14, 100 x 1 kb queries."""
    c = Case("c1", 31, 14, 3, 33, 200)
    G, L = 10, 5_000_000
    for g in range(G):
        c.genome_files.append((f"genome{g}.fa", fasta(f"genome{g}", genome_bases(g, 0, L)), False))
    _std_queries(c, G, L, 100)
    # h=14 gives empty hit lists for 1 kb queries: add long ones
    for q in range(100, 104):
        g, off = query_origin(q, G, L, 50_000)
        c.queries.append((f">long{q}_g{g}_p{off}".encode(), genome_bases(g, off, 50_000)))
    return c


def case_c2mini() -> Case:
    """BASELINE config 2's regime in small: 5 Mb genomes at -k 31 -h 17 have FULL sketches
    (sketch_size = 2^17 exactly), so active * active wraps to 0 in the reference's u32 arithmetic
    (Miekki.cpp:289, 306), every genome_size is 0, every intersection estimate is 0 and the
    approximate mode prints empty hit lists although the scores are right."""
    c = Case("c2mini", 31, 17, 3, 33, 200)
    G, L = 3, 5_000_000
    for g in range(G):
        c.genome_files.append((f"genome{g}.fa", fasta(f"genome{g}", genome_bases(g, 0, L)), False))
    _std_queries(c, G, L, 30)
    c.queries.append((b">miss", genome_bases(4_000_000, 0, 1000)))
    g, off = query_origin(31, G, L, 50_000)
    c.queries.append((b">long31", genome_bases(g, off, 50_000)))
    return c


def case_h16z() -> Case:
    """-h 16 with genomes long enough for a full sketch (65536^2 wraps to exactly 0 as well) next to
    a short one whose sketch is partial (non-zero genome_size): both regimes in one index."""
    c = Case("h16z", 31, 16, 3, 33, 20)
    lens = [3_200_000, 1_000_000, 150_000, 40_000]
    for g, n in enumerate(lens):
        c.genome_files.append((f"z{g}.fa", fasta(f"z{g}", genome_bases(700 + g, 0, n)), g == 1))
    for q in range(24):
        g = q % 4
        off = splitmix64_int(SEED_Q ^ (11000 + q)) % (lens[g] - 3000)
        c.queries.append((f">z{q}_g{g}".encode(), genome_bases(700 + g, off, 600 + 100 * (q % 9))))
    c.queries.append((b">whole_small", genome_bases(703, 0, 40_000)))
    c.queries.append((b">miss", genome_bases(4_100_000, 0, 1500)))
    return c


def case_h20() -> Case:
    """h=20 with every awkward query shape the reference accepts."""
    c = Case("h20", 31, 20, 3, 32, 200)
    G, L = 6, 1_000_000
    for g in range(G):
        c.genome_files.append((f"genome{g}.fa", fasta(f"genome{g}", genome_bases(g, 0, L)), g == 2))
    _std_queries(c, G, L, 40)
    for q in range(40, 50):                                  # non-genomic queries
        c.queries.append((f">miss{q}".encode(), genome_bases(1_000_000 + q, 0, 1000)))
    for q in range(50, 60):                                  # 1 % substitutions
        g, off = query_origin(q, G, L, 1000)
        c.queries.append((f">mut{q}_g{g}".encode(), mutate(genome_bases(g, off, 1000), q, 0.01)))
    g, off = query_origin(60, G, L, 50_000)
    c.queries.append((b">long60", genome_bases(g, off, 50_000)))
    c.queries.append((b">whole_genome3", genome_bases(3, 0, L)))
    s = bytearray(genome_bases(1, 1234, 1000))               # N and lower-case inside
    s[100] = ord("N"); s[101] = ord("N"); s[500:520] = bytes(s[500:520]).lower(); s[700] = ord("n")
    c.queries.append((b">with_N_and_lower", bytes(s)))
    s = bytearray(genome_bases(1, 4321, 1000)); s[3] = ord("N")  # non-ACGT inside the k-1 seed
    c.queries.append((b">N_in_seed", bytes(s)))
    c.queries.append((b">len_k", genome_bases(0, 10, 31)))       # zero k-mers processed
    c.queries.append((b">len_k_plus_1", genome_bases(0, 10, 32)))
    c.queries.append((b">len_k_minus_1", genome_bases(0, 10, 30)))  # skipped by the driver
    c.queries.append((b">len_k_plus_5", genome_bases(0, 777, 36)))
    return c


def case_w16() -> Case:
    """2-byte fingerprints: reference built with minimizer=uint16_t, run with -f 11."""
    c = Case("w16", 31, 17, 11, 33, 100)
    G, L = 6, 500_000
    for g in range(G):
        c.genome_files.append((f"genome{g}.fa", fasta(f"genome{g}", genome_bases(g, 0, L)), False))
    _std_queries(c, G, L, 30)
    for q in range(30, 36):
        c.queries.append((f">miss{q}".encode(), genome_bases(2_000_000 + q, 0, 1000)))
    g, off = query_origin(36, G, L, 20_000)
    c.queries.append((b">long36", genome_bases(g, off, 20_000)))
    return c


def case_messy() -> Case:
    """k=21, h=12: multi-FASTA contigs, ragged/empty lines, lower-case and N in
    genomes, a gzip'd genome, a too-short genome, many buckets shared per query."""
    c = Case("messy", 21, 12, 3, 32, 20)
    L = 60_000
    a = genome_bases(50, 0, L)
    c.genome_files.append(("multi.fa", b">c1 first\n" + a[:20_000] + b"\n\n>c2\n" + a[20_000:20_010]
                           + b"\n" + a[20_010:45_000] + b"\n>c3\n" + a[45_000:] + b"\n", False))
    bseq = bytearray(genome_bases(51, 0, L))
    bseq[1000:1100] = bytes(bseq[1000:1100]).lower()
    for p in (5, 3000, 3001, 3002, 40_000):
        bseq[p] = ord("N")
    c.genome_files.append(("lowerN.fa", fasta("lowerN", bytes(bseq), 70), True))
    c.genome_files.append(("tiny.fa", b">tiny\nACGTACGT\n", False))          # < k: skipped
    c.genome_files.append(("plain.fa", fasta("plain", genome_bases(52, 0, L), 61), False))
    c.genome_files.append(("nohdr.fa", genome_bases(53, 0, 30_000) + b"\n", False))
    srcs = [a, bytes(bseq), genome_bases(52, 0, L), genome_bases(53, 0, 30_000)]
    for q in range(24):
        src = srcs[q % 4]
        off = splitmix64_int(SEED_Q ^ (7000 + q)) % (len(src) - 3000)
        n = 300 + 100 * (q % 20)
        c.queries.append((f">m{q}".encode(), src[off:off + n]))
    c.queries.append((b">whole_multi", a))
    c.queries.append((b">miss", genome_bases(3_000_000, 0, 2500)))
    return c


def case_flush() -> Case:
    """Exact mode with more than 100 pending queries per genome file: the reference verifies a
    file's queries as soon as 100 have accumulated (Miekki.cpp:745-748) and the rest at the end in
    unordered_map order (752-754), which fixes the line order of the output."""
    c = Case("flush", 21, 12, 3, 32, 20)
    L = 40_000
    for g in range(4):
        c.genome_files.append((f"fl{g}.fa", fasta(f"fl{g}", genome_bases(300 + g, 0, L), 60), g == 1))
    for q in range(330):
        g = 0 if q % 10 < 7 else 1 + q % 3                     # 231 queries on fl0: two flushes + a rest
        off = splitmix64_int(SEED_Q ^ (9000 + q)) % (L - 700)
        c.queries.append((f">f{q}_g{g}".encode(), genome_bases(300 + g, off, 400 + 3 * (q % 90))))
    return c


def case_dups() -> Case:
    """Three hundred copies of one genome (the same file listed again and again) among a few others: every copy
    ties with every other, ties REPLACE in the reference's heap (Miekki.cpp:387-391), so each copy is a heap
    entrant -- far more than an entrant row holds on any shard.  Pins the overflow replays (dense score rows ->
    the host heap) of the single-context and of the sharded paths to the reference's own tie order."""
    c = Case("dups", 21, 12, 3, 32, 20)
    L = 30_000
    c.genome_files.append(("other0.fa", fasta("other0", genome_bases(810, 0, L)), False))
    for i in range(300):
        c.genome_files.append(("dup.fa", fasta("dup", genome_bases(800, 0, L)), False))
        if i % 97 == 50:
            c.genome_files.append((f"other{i}.fa", fasta(f"other{i}", genome_bases(811 + i, 0, L)), False))
    for q in range(12):
        src = 800 if q % 3 else 810
        off = splitmix64_int(SEED_Q ^ (12000 + q)) % (L - 2500)
        c.queries.append((f">d{q}".encode(), genome_bases(src, off, 500 + 150 * q)))
    c.queries.append((b">whole_dup", genome_bases(800, 0, L)))
    return c


RND_PARAMS = [(15, 10, 3, 32, 5), (27, 16, 11, 33, 30), (31, 13, 3, 33, 0), (9, 8, 3, 32, 12),
              (31, 20, 3, 33, 100), (21, 17, 11, 32, 50)]   # k, h, f, b, threshold


def case_rnd(i: int) -> Case:
    """Seeded random case: odd parameters, genomes of every awkward shape (exactly k and k+1 long,
    N / lower case / junk characters, a period-97 repeat, multi-FASTA, gzip, a header-only file) and
    a query file that mixes all sketch paths: short, shorter than k, exactly k, runs of long reads,
    a repetitive long one, a whole genome, a record without sequence."""
    import numpy as np
    k, h, f, b, thr = RND_PARAMS[i]
    c = Case(f"rnd{i}", k, h, f, b, thr)
    rng = np.random.default_rng(4242 + i)

    def messy(seq):
        s = bytearray(seq)
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, len(s))); n = int(rng.integers(1, 30))
            s[p:p + n] = bytes(rng.choice(np.frombuffer(b"Nnacgtxy", np.uint8), min(n, len(s) - p)))
        return bytes(s)

    raws = []
    # (no genome of exactly k characters: its sketch_size is 0, min_score 0 then makes 0/0 = NaN
    # intersections, and the reference is built with -Ofast, i.e. finite-math-only: what its heap
    # does with NaN is whatever that compiler run emitted -- nothing an oracle can pin)
    lens = [k + 2, k + 1, 300, 5000, 20_000, 60_000, 20_000, k - 1, 5000]
    for g, n in enumerate(lens):
        base = genome_bases(6000 + 40 * i + g, 0, n)
        if g == 4:
            base = (base[:97] * (n // 97 + 1))[:n]
        base = messy(base) if n > 100 else base
        raws.append(base)
        if g == 5:                                              # multi-FASTA, blank line inside
            data = b">a\n" + base[:25_000] + b"\n\n>b second contig\n" + base[25_000:] + b"\n"
        elif g == 6:
            data = base + b"\n"                                 # no header at all
        else:
            data = fasta(f"r{i}_{g}", base, int(rng.choice([60, 80, 100000])))
        c.genome_files.append((f"r{i}_{g}.fa", data, g in (3, 6)))
    c.genome_files.append((f"r{i}_hdr.fa", b">only a header\n", False))
    big = raws[5]
    for q in range(14):
        src = raws[int(rng.choice([3, 4, 5, 6, 8]))]
        off = int(rng.integers(0, len(src) - 2200))
        c.queries.append((f">s{q}".encode(), messy(src[off:off + 300 + 130 * q])))
    c.queries.append((b">shorter_than_k", big[:k - 1]))
    c.queries.append((b">exactly_k", big[7:7 + k]))
    c.queries.append((b">k_plus_1", big[9:9 + k + 1]))
    long_len = 4096 + k + 777
    for q in range(3):                                          # a run of long reads
        c.queries.append((f">long{q}".encode(), big[1000 * q:1000 * q + long_len + 211 * q]))
    c.queries.append((b">miss", genome_bases(9_000_000 + i, 0, 800)))
    c.queries.append((b">repetitive_long", (b"ACGTTGCAT" * 1500)[:long_len]))
    c.queries.append((b">long_again", messy(raws[6][:long_len + 5])))
    c.queries.append((b">whole", big))
    c.queries.append((b">no_sequence", b""))
    return c


CASES = {"c1": case_c1, "c2mini": case_c2mini, "h16z": case_h16z, "h20": case_h20, "w16": case_w16, "messy": case_messy,
         "rnd0": lambda: case_rnd(0), "rnd1": lambda: case_rnd(1), "rnd2": lambda: case_rnd(2), "rnd3": lambda: case_rnd(3),
         "rnd4": lambda: case_rnd(4), "rnd5": lambda: case_rnd(5)}
EXTRA_CASES = {"flush": case_flush, "dups": case_dups}
# cases whose reference-WRITTEN index file (the CLI's -d output, bytes as the reference's zstr
# writer made them) is committed as tests/golden/<name>_ref_idx.gz, for the -i loaders
REF_INDEX_CASES = ("rnd3",)       # exact-mode-only fixtures (tests/golden/<name>_exact.txt)
