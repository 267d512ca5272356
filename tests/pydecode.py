"""A third, deliberately naive restatement of the DINT decode in pure Python.

Used only to cross-check the hand-assembled vectors in tests/golden (whose
expected values were written down without running any decoder) and the C oracle
on small inputs. Follows the byte formats of SURVEY.md Appendix A.
"""
import struct


def parse_single_packed(b):
    m_size, n_off, n_tab = struct.unpack_from("<3I", b, 0)
    offsets = struct.unpack_from("<%dI" % n_off, b, 12)
    table = struct.unpack_from("<%dI" % n_tab, b, 12 + 4 * n_off)

    def entry(_d, i):
        size, off = (offsets[i] >> 24) + 1, offsets[i] & 0xFFFFFF
        return [table[off + k] if size <= 16 else 0 for k in range(size)]

    return entry


def parse_rectangular(b):
    (m_size,) = struct.unpack_from("<I", b, 0)
    rows = struct.unpack_from("<%dI" % (m_size * 17), b, 4)

    def entry(_d, i):
        size = rows[i * 17 + 16]
        return [rows[i * 17 + k] if size <= 16 else 0 for k in range(size)]

    return entry


def parse_multi_packed(b):
    m_size, n_start, n_off, n_tab = struct.unpack_from("<4I", b, 0)
    start = struct.unpack_from("<%dI" % n_start, b, 16)
    offsets = struct.unpack_from("<%dI" % n_off, b, 16 + 4 * n_start)
    table = struct.unpack_from("<%dI" % n_tab, b, 16 + 4 * n_start + 4 * n_off)

    def entry(d, i):
        o = offsets[start[d] + i]
        size, off = (o >> 24) + 1, o & 0xFFFFFF
        return [table[off + k] if size <= 16 else 0 for k in range(size)]

    return entry


def decode_slots(entry, data, pos, n, width, d):
    """Decode codewords until n integers are out. -> (ints, new pos)"""
    out = []
    step = width // 8
    while len(out) < n:
        idx = data[pos] if width == 8 else data[pos] | (data[pos + 1] << 8)
        pos += step
        if idx >= 2:
            out += entry(d, idx)
        elif idx == 1:
            out.append(struct.unpack_from("<I", data, pos)[0])
            pos += 4
        else:
            out.append(struct.unpack_from("<H", data, pos)[0])
            pos += 2
    assert len(out) == n, "codeword overshoots n"
    return out, pos


def decode_single(entry, data, pos, n):
    return decode_slots(entry, data, pos, n, 16, 0)


def decode_multi(entry, data, pos, n):
    out = []
    while len(out) < n:
        size = min(256, n - len(out))
        sel = data[pos]
        pos += 1
        width, d = (16, sel) if sel < 6 else (8, sel - 6)
        ints, pos = decode_slots(entry, data, pos, size, width, d)
        out += ints
    return out, pos


# ---- in-index blocks (include/ds2i/interpolative_coding.hpp:79-146, block_codecs.hpp:130-150) -------------------
class _Bits:
    def __init__(self, data, pos):
        self.data, self.byte, self.bit = data, pos, 0

    def read(self, n):
        v = 0
        for k in range(n):
            at = self.bit + k
            b = self.data[self.byte + (at >> 3)] if self.byte + (at >> 3) < len(self.data) else 0
            v |= ((b >> (at & 7)) & 1) << k
        self.bit += n
        return v

    def read_int(self, u):
        b = u.bit_length() - 1
        m = (1 << (b + 1)) - u
        v = self.read(b)
        if v >= m:
            v = (v << 1) + self.read(1) - m
        return v


def _interp(bits, out, lo_i, n, low, high):
    if n == 0:
        return
    h = n // 2
    val = low + bits.read_int(high - low + 1)
    out[lo_i + h] = val
    _interp(bits, out, lo_i, h, low, val)
    _interp(bits, out, lo_i + h + 1, n - h - 1, val, high)


def decode_interpolative(data, pos, sum_of_values, n):
    """interpolative_block::decode -> (gaps, new pos)"""
    if sum_of_values == 0xFFFFFFFF:
        sum_of_values, shift = 0, 0
        while True:
            c = data[pos]
            pos += 1
            sum_of_values += (c & 127) << shift
            shift += 7
            if c & 128:
                break
    out = [0] * n
    out[n - 1] = sum_of_values
    used = 0
    if n > 1:
        bits = _Bits(data, pos)
        _interp(bits, out, 0, n - 1, 0, sum_of_values)
        used = (bits.bit + 7) // 8
        for i in range(n - 1, 0, -1):
            out[i] -= out[i - 1]
    return out, pos + used


def decode_posting_list(entry_docs, entry_freqs, data, pos, multi):
    """dict_posting_list::document_enumerator walked front to back (dict_posting_list.hpp:93-318) -> (docids, freqs)"""
    n, shift = 0, 0
    while True:
        c = data[pos]
        pos += 1
        n += (c & 127) << shift
        shift += 7
        if c & 128:
            break
    blocks = (n + 255) // 256
    maxs = struct.unpack_from("<%dI" % blocks, data, pos)
    ends = (0,) + struct.unpack_from("<%dI" % (blocks - 1), data, pos + 4 * blocks)
    base_at = pos + 4 * blocks + 4 * (blocks - 1)
    docids, freqs = [], []
    for b in range(blocks):
        size = 256 if (b + 1) * 256 <= n else n % 256
        cur_base = (maxs[b - 1] + 1) if b else 0
        at = base_at + ends[b]

        def part(entry, at, sum_of_values):
            if size < 256:
                return decode_interpolative(data, at, sum_of_values, size)
            if not multi:
                return decode_slots(entry, data, at, size, 16, 0)
            sel = data[at]
            width, d = (16, sel) if sel < 6 else (8, sel - 6)
            return decode_slots(entry, data, at + 1, size, width, d)

        gaps, at = part(entry_docs, at, (maxs[b] - cur_base - (size - 1)) & 0xFFFFFFFF)
        doc = cur_base + gaps[0]
        docids.append(doc)
        for g in gaps[1:]:
            doc += g + 1
            docids.append(doc)
        f, at = part(entry_freqs, at, 0xFFFFFFFF)
        freqs += [(v + 1) & 0xFFFFFFFF for v in f]
    return docids, freqs
