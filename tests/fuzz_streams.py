"""Differential fuzzing of the DINT decoders over their whole input space (test infrastructure).

Every large parity test decodes streams an ENCODER made (optimal / greedy parse over a DSF dictionary built from the
same data). The reference decoders accept any slot sequence over any dictionary file (vroom_env/dint_codecs.hpp:45-100,
:536-612; include/dint/dint_codecs.hpp:21-46): this module makes seeded random DICTIONARY FILES for the three formats
(single_dictionary.hpp:72-86, rectangular_dictionary.hpp:72-77, multi_dictionary.hpp:70-91) and random DECODER-LEGAL
slot streams assembled slot by slot — no encoder, no decoder is run to make them — and the integers they must decode
to, by substituting every codeword with its definition (a fourth statement of the format beside the HIP kernels, the C
oracle and tests/pydecode.py).

Domain (what "decoder-legal" means here):
 * dictionary files a builder's write() can emit: the reserved codewords as init() writes them (0/1 the exception
   markers, 2..6 the zero runs 256..16 at offset 0 of a table that starts with 16 zeros), every other entry 1..16
   integers of ANY value, at ANY table offset (duplicates, entries nested inside other entries, entries overlapping the
   leading zeros), m_size from 8 to 65536; multi: six contexts, any of them possibly holding its reserved codewords only;
 * streams: any sequence of slots whose sizes sum to exactly n (single: per list; multi / in-index: per 256-integer
   block), codeword ids below the dictionary's size (8-bit blocks: below 256), literals of any value — 32-bit
   exceptions carrying small values, halves equal to 0 / 1 / the run ids included.
Outside it (an entry of more than 16 integers, reserved rows that are not init()'s, ids past the dictionary) the reference
reads whatever lies behind its tables or what the previous list left in its buffer: undefined there, rejected here
(DINT_ERR_FORMAT) or simply never generated.
"""
import hashlib
import struct

import numpy as np

RECT, SINGLE, MULTI = 0, 1, 2
RUNS = {2: 256, 3: 128, 4: 64, 5: 32, 6: 16}
RESERVED = 7
ZEROS = 256  # the internal flat table starts with 256 zeros: runs are slices of it like any entry


def vbyte(v):
    """TightVariableByte::encode_single (vroom_env/codecs.hpp:75-91): 7 bits a byte, the LAST byte has bit 7 set."""
    out = bytearray()
    while v >= 128:
        out.append(v & 127)
        v >>= 7
    out.append(v | 128)
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------------
# dictionaries
# ---------------------------------------------------------------------------------------------------------------
VALUE_PROFILES = ("tiny", "byte_edge", "u16_edge", "wide", "mixed", "zeros")
SIZE_PROFILES = ("pow2", "any", "long", "short", "sixteen")


def _values(r, profile, count):
    if profile == "tiny":
        return r.integers(0, 4, count)
    if profile == "zeros":
        return np.where(r.random(count) < 0.8, 0, r.integers(0, 3, count))
    if profile == "byte_edge":
        return r.integers(250, 262, count)
    if profile == "u16_edge":
        return r.integers(65530, 65542, count)
    if profile == "wide":
        return r.integers(0, 1 << 32, count)
    # mixed: mostly small, some of everything
    pick = r.random(count)
    v = r.integers(0, 40, count)
    v = np.where(pick > 0.70, r.integers(250, 262, count), v)
    v = np.where(pick > 0.85, r.integers(65530, 65542, count), v)
    v = np.where(pick > 0.95, r.integers(0, 1 << 32, count), v)
    v = np.where(pick > 0.99, 0xFFFFFFFF, v)
    return v


def _sizes(r, profile, count):
    if profile == "pow2":
        return r.choice([1, 2, 4, 8, 16], count, p=[0.15, 0.25, 0.3, 0.2, 0.1])
    if profile == "any":
        return r.integers(1, 17, count)
    if profile == "long":
        return r.choice([8, 16, 15, 14, 7], count, p=[0.3, 0.5, 0.1, 0.05, 0.05])
    if profile == "short":
        return r.choice([1, 2, 3], count, p=[0.6, 0.3, 0.1])
    return np.full(count, 16)


class FuzzDictionary:
    """A dictionary file and the model of it the generator substitutes from.

    table   u32[]: ZEROS zeros, then the payload words (the file's table without its 16 leading zeros follows at ZEROS)
    size/off[d]  : per codeword id of context d — integers it decodes to, first word in `table`
    """

    def __init__(self, kind, file, table, size, off):
        self.kind, self.file, self.table, self.size, self.off = kind, file, table, size, off
        self.num_dicts = len(size)
        # candidates by size, for the filler and the size-targeted profiles
        self.by_size = []
        for d in range(self.num_dicts):
            m = {}
            ids = np.arange(RESERVED, len(size[d]))
            for s in np.unique(size[d][RESERVED:]) if len(ids) else []:
                m[int(s)] = ids[size[d][RESERVED:] == s]
            self.by_size.append(m)

    def entry(self, d, i):
        return self.table[self.off[d][i]: self.off[d][i] + self.size[d][i]]


def _context(r, count, value_profile, size_profile, payload, nest_p, dup_p):
    """`count` entries appended to the shared payload (list of u32 arrays: the file's table behind its 16 zeros).
    -> (sizes, file offsets; -1 = a nested entry, placed when the whole table is known)"""
    sizes = _sizes(r, size_profile, count).astype(np.int64)
    offs = np.full(count, -1, dtype=np.int64)
    kind = r.random(count)
    fresh = kind >= nest_p + dup_p
    if count:
        fresh[0] = True
    have = 16 + sum(len(p) for p in payload)
    fs = sizes[fresh]
    offs[fresh] = have + np.cumsum(fs) - fs
    words = _values(r, value_profile, int(fs.sum())).astype(np.uint64).astype(np.uint32)
    payload.append(words)
    words_at = have
    have += len(words)
    fresh_ids = np.flatnonzero(fresh)
    for i in np.flatnonzero((kind >= nest_p) & ~fresh):  # the same integers again, stored again
        j = int(fresh_ids[r.integers(0, len(fresh_ids))])
        at = int(offs[j]) - words_at
        sizes[i] = sizes[j]
        offs[i] = have
        payload.append(words[at: at + int(sizes[j])].copy())
        have += int(sizes[j])
    return sizes, offs


def make_dictionary(r, kind, m_entries, value_profile="mixed", size_profile="pow2", nest_p=0.15, dup_p=0.05,
                    context_entries=None):
    """A random dictionary file of `m_entries` codewords (reserved included; multi: per context, or the list
    `context_entries`, 7 = a context holding its reserved codewords only)."""
    nd = 6 if kind == MULTI else 1
    per = list(context_entries) if context_entries is not None else [m_entries] * nd
    assert len(per) == nd and all(RESERVED <= m <= 65536 for m in per)
    payload = []  # the file's table behind its 16 leading zeros
    sizes, offs = [], []
    for d in range(nd):
        vp = value_profile if isinstance(value_profile, str) else value_profile[d % len(value_profile)]
        s, o = _context(r, per[d] - RESERVED, vp, size_profile, payload, nest_p, dup_p)
        sizes.append(np.concatenate([[1, 1, 256, 128, 64, 32, 16], s]).astype(np.int64))
        offs.append(np.concatenate([np.zeros(RESERVED, dtype=np.int64), o]))
    payload = np.concatenate(payload) if payload else np.zeros(0, dtype=np.uint32)
    for d in range(nd):
        # nested entries: a sub-interval of the table anywhere — inside other entries, across their borders, overlapping
        # the leading zeros (the packed builders nest short entries inside long ones: single_dictionary.hpp:109-160)
        nested = offs[d] < 0
        limit = 16 + len(payload) - sizes[d][nested] + 1
        shrink = limit <= 0
        if shrink.any():  # a table shorter than the entry: the entry becomes its first integer's worth of zeros
            sizes[d][np.flatnonzero(nested)[shrink]] = 1
            limit = np.maximum(limit, 16)
        offs[d][nested] = (r.random(int(nested.sum())) * limit).astype(np.int64)
    file_table = np.concatenate([np.zeros(16, dtype=np.uint32), payload])
    # the generator's flat table: ZEROS zeros, then the file's payload; an entry that overlaps the file's leading zeros
    # keeps its place relative to the payload (file offset f -> ZEROS - 16 + f)
    table = np.concatenate([np.zeros(ZEROS - 16, dtype=np.uint32), file_table])
    model_off = []
    for d in range(nd):
        o = offs[d] + (ZEROS - 16)
        o[:RESERVED] = 0
        model_off.append(o)

    def packed_offsets(d):
        return (((sizes[d] - 1) << 24) | offs[d]).astype(np.uint32)

    if kind == SINGLE:
        o = packed_offsets(0)
        file = struct.pack("<3I", len(o), len(o), len(file_table)) + o.tobytes() + file_table.tobytes()
    elif kind == MULTI:
        starts = np.cumsum([0] + [len(s) for s in sizes[:-1]]).astype(np.uint32)
        o = np.concatenate([packed_offsets(d) for d in range(nd)])
        m_size = RESERVED + sum(len(s) - RESERVED for s in sizes)  # multi_dictionary.hpp:40,136: one global count
        file = (struct.pack("<4I", m_size, 6, len(o), len(file_table)) + starts.tobytes() + o.tobytes()
                + file_table.tobytes())
    else:
        m = len(sizes[0])
        rows = np.zeros((m, 17), dtype=np.uint32)
        rows[:, 16] = sizes[0]
        for i in range(RESERVED, m):
            s = int(sizes[0][i])
            rows[i, :s] = table[model_off[0][i]: model_off[0][i] + s]
        file = struct.pack("<I", m) + rows.tobytes()
    return FuzzDictionary(kind, file, table, sizes, model_off)


# ---------------------------------------------------------------------------------------------------------------
# slot sequences
# ---------------------------------------------------------------------------------------------------------------
STREAM_PROFILES = ("uniform", "cold16", "exceptions", "runs", "marker_literals", "low_ids", "high_ids", "small_e32",
                   "ones")
_LITERAL_IDS = np.array([0, 1, 2, 3, 4, 5, 6, 7, 255, 256, 65535], dtype=np.int64)


def _literal16(r, profile, count):
    if profile == "marker_literals":
        return r.choice(_LITERAL_IDS, count)
    v = r.integers(0, 65536, count)
    return np.where(r.random(count) < 0.3, r.choice(_LITERAL_IDS, count), v)


def _literal32(r, profile, count):
    if profile == "marker_literals":  # both halves look like markers / run ids
        return r.choice(_LITERAL_IDS[:8], count) | (r.choice(_LITERAL_IDS[:8], count) << 16)
    if profile == "small_e32":  # a 32-bit exception carrying what a 16-bit one could
        return r.choice(_LITERAL_IDS, count)
    v = r.integers(0, 1 << 32, count)
    pick = r.random(count)
    v = np.where(pick < 0.25, r.integers(65536, 65536 + 8, count), v)
    v = np.where(pick < 0.10, 0xFFFFFFFF, v)
    return v


def random_slots(r, D, d, n, narrow=False, profile="uniform"):
    """Slots of context d that decode to EXACTLY n integers.
    -> (ids i64[], literals i64[] (the value behind an exception marker, else 0))"""
    limit = min(len(D.size[d]), 256 if narrow else 65536)
    size = D.size[d]
    ids_out, lits_out = [], []
    left = n
    while left > 0:
        guess = max(4, min(left, 4000))
        if profile == "cold16" and 16 in D.by_size[d] and D.by_size[d][16][-1] < limit:
            c = D.by_size[d][16]
            c = c[c < limit]
            ids = c[r.integers(max(0, len(c) - 64), len(c), guess)]  # the LAST size-16 entries: the coldest
        elif profile == "exceptions":
            ids = np.where(r.random(guess) < 0.7, r.integers(0, 2, guess), r.integers(0, limit, guess))
        elif profile == "runs":
            ids = np.where(r.random(guess) < 0.6, r.integers(2, 7, guess), r.integers(0, limit, guess))
        elif profile in ("marker_literals", "small_e32"):
            ids = np.where(r.random(guess) < 0.5, r.integers(0, 2, guess), r.integers(0, limit, guess))
        elif profile == "low_ids":
            ids = r.integers(0, min(limit, 64), guess)
        elif profile == "high_ids":
            ids = r.integers(max(0, limit - 300), limit, guess)
        elif profile == "ones" and 1 in D.by_size[d] and D.by_size[d][1][0] < limit:
            c = D.by_size[d][1]
            ids = c[c < limit][r.integers(0, np.count_nonzero(c < limit), guess)]
        else:
            ids = r.integers(0, limit, guess)
            # a dictionary of 65536 entries: keep some weight on the reserved codewords
            ids = np.where(r.random(guess) < 0.08, r.integers(0, RESERVED, guess), ids)
        cum = np.cumsum(size[ids])
        keep = int(np.searchsorted(cum, left, side="right"))
        if keep:
            ids = ids[:keep]
            is16, is32 = ids == 0, ids == 1
            lits = np.zeros(keep, dtype=np.int64)
            lits[is16] = _literal16(r, profile, int(is16.sum()))
            lits[is32] = _literal32(r, profile, int(is32.sum()))
            ids_out.append(ids)
            lits_out.append(lits)
            left -= int(cum[keep - 1])
            if keep == guess or left == 0:
                continue
        # the next drawn slot overshoots: fill what is left of THIS stretch with sizes that fit, largest first at random
        fit = [s for s in D.by_size[d] if s <= left and D.by_size[d][s][0] < limit]
        runs = [i for i, s in RUNS.items() if s <= left]
        choice = r.random()
        if fit and choice < 0.6:
            s = fit[int(r.integers(0, len(fit)))] if choice < 0.2 else max(fit)
            c = D.by_size[d][s]
            c = c[c < limit]
            pick = int(c[r.integers(0, len(c))])
            lit = 0
        elif runs and choice < 0.8:
            pick, lit = runs[int(r.integers(0, len(runs)))], 0
        else:
            pick = int(r.integers(0, 2))
            lit = int((_literal16 if pick == 0 else _literal32)(r, profile, 1)[0])
        ids_out.append(np.array([pick], dtype=np.int64))
        lits_out.append(np.array([lit], dtype=np.int64))
        left -= int(size[pick])
    if not ids_out:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return np.concatenate(ids_out), np.concatenate(lits_out)


def expand(D, d, ids, lits):
    """What the slots decode to, by substitution: codeword -> its entry, marker -> its literal."""
    sizes = D.size[d][ids]
    total = int(sizes.sum())
    first = np.cumsum(sizes) - sizes
    src = np.repeat(D.off[d][ids] - first, sizes) + np.arange(total)
    out = D.table[src]
    exc = ids < 2
    out[first[exc]] = lits[exc].astype(np.uint32)
    return out


def slot_bytes(ids, lits, narrow=False):
    """The slots as stream bytes -> (u8[], byte position of every slot + the end)."""
    w = 1 if narrow else 2
    lens = np.full(len(ids), w, dtype=np.int64)
    lens[ids == 0] += 2
    lens[ids == 1] += 4
    pos = np.concatenate([[0], np.cumsum(lens)])
    buf = np.zeros(int(pos[-1]), dtype=np.uint8)
    at = pos[:-1]
    buf[at] = ids & 0xFF
    if not narrow:
        buf[at + 1] = ids >> 8
    exc = ids < 2
    for k in range(2):
        buf[at[exc] + w + k] = (lits[exc] >> (8 * k)) & 0xFF
    e32 = ids == 1
    for k in range(2, 4):
        buf[at[e32] + w + k] = (lits[e32] >> (8 * k)) & 0xFF
    return buf, pos


# ---------------------------------------------------------------------------------------------------------------
# vroom streams (vroom_env/jobs.hpp:89-91: { vbyte(n) vbyte(universe) payload }*)
# ---------------------------------------------------------------------------------------------------------------
UNIT_DTYPE = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("n", "<u4"), ("list", "<u4")])


class FuzzStream:
    """enc u8[], expect u32[], lists [(payload offset, n, first integer)], units UNIT_DTYPE[] cut at random legal
    boundaries, ends u64[] — the byte offset behind every unit's last slot."""

    def __init__(self, enc, expect, lists, units, ends):
        self.enc, self.expect, self.lists, self.units, self.ends = enc, expect, lists, units, ends


def _cut(r, boundaries_int, boundaries_byte, n, cut_p):
    """Units of one list: cuts at some of the given (integer position, byte position) boundaries."""
    inner = np.arange(1, len(boundaries_int) - 1)
    if len(inner) and cut_p > 0:
        take = inner[r.random(len(inner)) < cut_p]
    else:
        take = inner[:0]
    # a unit must hold at least one integer: boundaries behind zero-size progress never occur (every slot has size >= 1)
    at = np.concatenate([[0], take, [len(boundaries_int) - 1]])
    return boundaries_int[at], boundaries_byte[at]


def make_stream(r, D, n_lists, max_n=3000, profiles=STREAM_PROFILES, cut_p=0.05):
    """A vroom stream of `n_lists` lists over dictionary D with its expected integers and a random unit table."""
    multi = D.kind == MULTI
    chunks, expect, lists, units, ends = [], [], [], [], []
    pos = out_pos = 0
    for li in range(n_lists):
        u = r.random()
        n = int(r.integers(1, 40)) if u < 0.35 else int(r.integers(1, 600)) if u < 0.8 else int(r.integers(1, max_n + 1))
        if r.random() < 0.15:
            n = int(r.choice([1, 15, 16, 17, 255, 256, 257, 511, 512, 513, 1024]))
        if li % 97 == 50:  # a few long lists: many tiles, tiles of more than 2048 integers (runs), units of any length
            n = int(r.integers(20_000, 120_000))
        header = vbyte(n) + vbyte(int(r.integers(0, 1 << int(r.integers(1, 33)))))
        profile = profiles[int(r.integers(0, len(profiles)))]
        payload_at = pos + len(header)
        parts = [np.frombuffer(header, dtype=np.uint8)]
        if not multi:
            ids, lits = random_slots(r, D, 0, n, False, profile)
            body, bpos = slot_bytes(ids, lits)
            expect.append(expand(D, 0, ids, lits))
            parts.append(body)
            sizes = D.size[0][ids]
            b_int = np.concatenate([[0], np.cumsum(sizes)])
            b_byte = bpos + payload_at
        else:
            b_int, b_byte, at = [0], [payload_at], payload_at
            for first in range(0, n, 256):
                bn = min(256, n - first)
                sel = int(r.integers(0, 12))
                narrow, d = sel >= 6, sel % 6
                ids, lits = random_slots(r, D, d, bn, narrow, profile)
                body, _ = slot_bytes(ids, lits, narrow)
                expect.append(expand(D, d, ids, lits))
                parts += [np.array([sel], dtype=np.uint8), body]
                at += 1 + len(body)
                b_int.append(first + bn)
                b_byte.append(at)
            b_int, b_byte = np.array(b_int), np.array(b_byte)
        ci, cb = _cut(r, b_int, b_byte, n, cut_p if r.random() < 0.7 else 0.5)
        for k in range(len(ci) - 1):
            units.append((int(cb[k]), out_pos + int(ci[k]), int(ci[k + 1] - ci[k]), li))
            ends.append(int(cb[k + 1]))
        lists.append((payload_at, n, out_pos))
        blob = np.concatenate(parts)
        chunks.append(blob)
        pos += len(blob)
        out_pos += n
    enc = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    exp = np.concatenate(expect) if expect else np.zeros(0, dtype=np.uint32)
    return FuzzStream(enc, exp, lists, np.array(units, dtype=UNIT_DTYPE), np.array(ends, dtype=np.uint64))


# ---------------------------------------------------------------------------------------------------------------
# the case list: (seed, kind, dictionary shape) -> FuzzDictionary + FuzzStream
# ---------------------------------------------------------------------------------------------------------------
def plan(n_dicts_per_kind, lists_per_dict):
    """The committed fuzz plan: a list of (seed, kind, kwargs for make_dictionary, lists)."""
    shapes = [
        dict(m_entries=8, value_profile="mixed", size_profile="pow2"),
        dict(m_entries=9, value_profile="wide", size_profile="sixteen"),
        dict(m_entries=40, value_profile="byte_edge", size_profile="any"),
        dict(m_entries=255, value_profile="mixed", size_profile="pow2"),
        dict(m_entries=256, value_profile="u16_edge", size_profile="long"),
        dict(m_entries=257, value_profile="tiny", size_profile="short"),
        dict(m_entries=3000, value_profile="mixed", size_profile="any", nest_p=0.4),
        dict(m_entries=20000, value_profile="tiny", size_profile="pow2", nest_p=0.5, dup_p=0.2),
        dict(m_entries=65536, value_profile="mixed", size_profile="pow2"),
        dict(m_entries=65536, value_profile="byte_edge", size_profile="long", nest_p=0.3),
        dict(m_entries=65535, value_profile="zeros", size_profile="any"),
        dict(m_entries=12000, value_profile="wide", size_profile="pow2", nest_p=0.0),
        dict(m_entries=65536, value_profile="tiny", size_profile="sixteen", nest_p=0.0, dup_p=0.0),
        dict(m_entries=30000, value_profile="u16_edge", size_profile="pow2"),
    ]
    out = []
    for kind in (SINGLE, RECT, MULTI):
        for k in range(n_dicts_per_kind):
            shape = dict(shapes[k % len(shapes)])
            if kind == MULTI:
                if k % 3 == 1:  # contexts of very different sizes, some holding their reserved codewords only
                    m = shape["m_entries"]
                    shape["context_entries"] = [m, 7, max(7, m // 100), 7, min(65536, m + 5), 300]
                shape["value_profile"] = (shape["value_profile"], "tiny", "mixed")
                if shape["m_entries"] > 30000 and k % 2:
                    shape["m_entries"] = 30000  # six contexts of 65536 long entries: keep the set-up short
            out.append((1000 * (kind + 1) + k, kind, shape, lists_per_dict))
    return out


def build_case(case):
    seed, kind, shape, n_lists = case
    r = np.random.default_rng(seed)
    D = make_dictionary(r, kind, **shape)
    S = make_stream(r, D, n_lists)
    return D, S


def digest(D, S):
    h = hashlib.sha256()
    for a in (np.frombuffer(D.file, dtype=np.uint8), S.enc, S.expect, S.units.view(np.uint8), S.ends):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------
# in-index posting lists (include/dint/dict_posting_list.hpp:10-56: vbyte(n) | max[B] | endpoint[B-1] | blocks)
# ---------------------------------------------------------------------------------------------------------------
class _BitWriter:
    """bit_writer (include/ds2i/interpolative_coding.hpp:10-77): LSB first."""

    def __init__(self):
        self.acc, self.bits = 0, 0

    def write(self, v, n):
        self.acc |= (v & ((1 << n) - 1)) << self.bits
        self.bits += n

    def write_int(self, val, u):
        b = u.bit_length() - 1
        m = (1 << (b + 1)) - u
        if val < m:
            self.write(val, b)
        else:
            val += m
            self.write(val >> 1, b)
            self.write(val & 1, 1)

    def interpolative(self, seq, lo, n, low, high):
        if not n:
            return
        h = n // 2
        val = seq[lo + h]
        self.write_int(val - low, high - low + 1)
        self.interpolative(seq, lo, h, low, val)
        self.interpolative(seq, lo + h + 1, n - h - 1, val, high)

    def bytes(self):
        return self.acc.to_bytes((self.bits + 7) // 8, "little")


def interpolative_block(values, sum_known):
    """interpolative_block::encode (include/ds2i/block_codecs.hpp:104-128); sum_known False: the vbyte sum first."""
    prefix = np.cumsum(np.asarray(values, dtype=np.int64))
    total = int(prefix[-1])
    assert total < 0xFFFFFFFF
    w = _BitWriter()
    w.interpolative([int(v) for v in prefix], 0, len(values) - 1, 0, total)
    return (b"" if sum_known else vbyte(total)) + w.bytes()


class FuzzIndex:
    """index u8[] (lists back to back), offsets u64[n_lists + 1], docids / freqs u32[] of all lists back to back,
    bounds: first posting of every list."""

    def __init__(self, index, offsets, docids, freqs, bounds):
        self.index, self.offsets, self.docids, self.freqs, self.bounds = index, offsets, docids, freqs, bounds


def _block_part(r, D, n, profile, lit_cap):
    """One full block's docs or freqs part: (selector byte +) slots of exactly 256 integers -> (bytes, integers)"""
    if D.kind == MULTI:
        sel = int(r.integers(0, 12))
        narrow, d = sel >= 6, sel % 6
    else:
        sel, narrow, d = None, False, 0
    ids, lits = random_slots(r, D, d, n, narrow, profile)
    lits = lits % lit_cap
    body, _ = slot_bytes(ids, lits, narrow)
    if sel is not None:
        body = np.concatenate([np.array([sel], dtype=np.uint8), body])
    return body, expand(D, d, ids, lits)


def make_index(r, Dd, Df, n_lists, max_n=1500, profiles=STREAM_PROFILES, value_cap=1 << 17):
    """Posting lists over a docs dictionary Dd (values below value_cap: docIDs of a list stay far below 2^32) and a freqs
    dictionary Df (any values)."""
    assert int(Dd.table.max(initial=0)) < value_cap
    lists, offsets, docids, freqs, bounds = [], [0], [], [], [0]
    for _ in range(n_lists):
        u = r.random()
        n = int(r.integers(1, 256)) if u < 0.45 else int(r.integers(256, max_n + 1))
        if r.random() < 0.2:
            n = int(r.choice([1, 2, 255, 256, 257, 511, 512, 513]))
        blocks = (n + 255) // 256
        profile = profiles[int(r.integers(0, len(profiles)))]
        maxs, ends, body = [], [], []
        size_so_far, base = 0, 0
        for b in range(blocks):
            size = min(256, n - 256 * b)
            if size == 256:
                dbytes, dvals = _block_part(r, Dd, 256, profile, value_cap)
                fbytes, fvals = _block_part(r, Df, 256, profile, 0xFFFFFFFF)
            else:
                dvals = np.where(r.random(size) < 0.5, 0, r.integers(0, 1 << int(r.integers(1, 17)), size)).astype(np.uint32)
                fvals = np.where(r.random(size) < 0.6, 0, r.integers(0, 1 << int(r.integers(1, 24)), size)).astype(np.uint32)
                dbytes = np.frombuffer(interpolative_block(dvals, True), dtype=np.uint8)
                fbytes = np.frombuffer(interpolative_block(fvals, False), dtype=np.uint8)
            ids = base + np.cumsum(dvals.astype(np.int64)) + np.arange(size)
            assert int(ids[-1]) < (1 << 32)
            docids.append(ids.astype(np.uint32))
            freqs.append((fvals.astype(np.int64) + 1).astype(np.uint32))  # (0xFFFFFFFF + 1 wraps like the u32 it is)
            maxs.append(int(ids[-1]))
            base = int(ids[-1]) + 1
            body += [dbytes, fbytes]
            size_so_far += len(dbytes) + len(fbytes)
            if b != blocks - 1:
                ends.append(size_so_far)
        head = vbyte(n) + struct.pack("<%dI" % blocks, *maxs) + struct.pack("<%dI" % (blocks - 1), *ends)
        blob = np.concatenate([np.frombuffer(head, dtype=np.uint8)] + body)
        lists.append(blob)
        offsets.append(offsets[-1] + len(blob))
        bounds.append(bounds[-1] + n)
    return FuzzIndex(np.concatenate(lists), np.array(offsets, dtype=np.uint64), np.concatenate(docids),
                     np.concatenate(freqs), np.array(bounds, dtype=np.int64))


def index_plan(n_per_kind, lists_each):
    """(seed, kind, docs dictionary shape, freqs dictionary shape, lists)"""
    docs_shapes = [
        dict(m_entries=8, value_profile="tiny", size_profile="pow2"),
        dict(m_entries=300, value_profile="byte_edge", size_profile="any"),
        dict(m_entries=5000, value_profile="u16_edge", size_profile="pow2", nest_p=0.4),
        dict(m_entries=65536, value_profile="tiny", size_profile="long", nest_p=0.3),
        dict(m_entries=65536, value_profile="u16_edge", size_profile="pow2"),
        dict(m_entries=256, value_profile="zeros", size_profile="sixteen"),
    ]
    freqs_shapes = [
        dict(m_entries=9, value_profile="wide", size_profile="pow2"),
        dict(m_entries=65536, value_profile="mixed", size_profile="any"),
        dict(m_entries=700, value_profile="tiny", size_profile="short"),
        dict(m_entries=8, value_profile="zeros", size_profile="pow2"),
        dict(m_entries=20000, value_profile="mixed", size_profile="long", nest_p=0.5),
        dict(m_entries=257, value_profile="byte_edge", size_profile="pow2"),
    ]
    out = []
    for kind in (SINGLE, RECT, MULTI):
        for k in range(n_per_kind):
            ds, fs = dict(docs_shapes[k % len(docs_shapes)]), dict(freqs_shapes[(k + kind) % len(freqs_shapes)])
            if kind == MULTI:
                for s in (ds, fs):
                    s["m_entries"] = min(s["m_entries"], 20000)
                if k % 2:
                    m = ds["m_entries"]
                    ds["context_entries"] = [m, 7, 300, 7, max(7, m // 10), 40]
            out.append((7000 * (kind + 1) + k, kind, ds, fs, lists_each))
    return out


def build_index_case(case):
    seed, kind, ds, fs, n_lists = case
    r = np.random.default_rng(seed)
    Dd = make_dictionary(r, kind, **ds)
    Df = make_dictionary(r, kind, **fs)
    return Dd, Df, make_index(r, Dd, Df, n_lists)


def index_digest(Dd, Df, X):
    h = hashlib.sha256()
    for a in (np.frombuffer(Dd.file, dtype=np.uint8), np.frombuffer(Df.file, dtype=np.uint8), X.index,
              X.offsets.view(np.uint8), X.docids, X.freqs):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]
