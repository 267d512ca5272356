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
