"""Tiny raw-BAM record builder/parser for crafted test cases (the reference builds its fixtures
programmatically with `SamBuilder` and commits no BAM files — SURVEY.md §4)."""
import struct

CODES = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def cigar_ops(cigar: str):
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | OPS[ch])
            num = ""
    return out


def make_record(name, seq, quals, flag=0, ref_id=0, pos=99, mapq=60, cigar=None, mate_ref=-1, mate_pos=-1, tlen=0, tags=()):
    """tags: sequence of (tag, type, value) with type in Z, i (smallest int), f, raw."""
    name_b = name.encode() + b"\0"
    ops = cigar_ops(cigar) if cigar is not None else ([(len(seq) << 4)] if not (flag & 0x4) else [])
    packed = bytearray()
    for i in range(0, len(seq), 2):
        hi = CODES.get(seq[i].upper(), 15)
        lo = CODES.get(seq[i + 1].upper(), 15) if i + 1 < len(seq) else 0
        packed.append((hi << 4) | lo)
    if quals is None:
        q = bytes([0xFF] * len(seq))
    else:
        q = bytes(quals)
    aux = bytearray()
    for tag, ty, val in tags:
        aux += tag.encode()
        if ty == "Z":
            aux += b"Z" + (val if isinstance(val, bytes) else val.encode()) + b"\0"
        elif ty == "i":
            if -128 <= val <= 127:
                aux += b"c" + struct.pack("<b", val)
            elif 0 <= val <= 255:
                aux += b"C" + struct.pack("<B", val)
            elif 0 <= val <= 65535:
                aux += b"S" + struct.pack("<H", val)
            elif -32768 <= val <= 32767:
                aux += b"s" + struct.pack("<h", val)
            else:
                aux += b"i" + struct.pack("<i", val)
        elif ty == "f":
            aux += b"f" + struct.pack("<f", val)
        elif ty == "raw":
            aux += val
    head = struct.pack("<iiBBHHHIiii", ref_id, pos, len(name_b), mapq, 4680, len(ops), flag, len(seq), mate_ref, mate_pos, tlen)
    return bytes(head + name_b + b"".join(struct.pack("<I", o) for o in ops) + bytes(packed) + q + bytes(aux))


def frag(name, seq, quals, mi, pos=99, cigar=None, flag=0, **kw):
    q = quals if not isinstance(quals, int) else [quals] * len(seq)
    return make_record(name, seq, q, flag=flag, pos=pos, cigar=cigar, tags=[("MI", "Z", mi)] + list(kw.get("tags", ())))


def pair(name, seq1, q1, seq2, q2, mi, pos1=99, pos2=None, cigar1=None, cigar2=None, rx=None, ref_id=0, extra=()):
    """FR pair: R1 forward at pos1, R2 reverse at pos2 (0-based); sequences in stored (reference) orientation."""
    L1, L2 = len(seq1), len(seq2)
    if pos2 is None:
        pos2 = pos1 + 50
    c1 = cigar1 or f"{L1}M"
    c2 = cigar2 or f"{L2}M"
    q1 = [q1] * L1 if isinstance(q1, int) else q1
    q2 = [q2] * L2 if isinstance(q2, int) else q2
    ins = pos2 + L2 - pos1
    tags = [("MI", "Z", mi)] + ([("RX", "Z", rx)] if rx else []) + list(extra)
    r1 = make_record(name, seq1, q1, flag=0x1 | 0x2 | 0x40 | 0x20, ref_id=ref_id, pos=pos1, cigar=c1, mate_ref=ref_id, mate_pos=pos2, tlen=ins,
                     tags=tags + [("MC", "Z", c2)])
    r2 = make_record(name, seq2, q2, flag=0x1 | 0x2 | 0x80 | 0x10, ref_id=ref_id, pos=pos2, cigar=c2, mate_ref=ref_id, mate_pos=pos1, tlen=-ins,
                     tags=tags + [("MC", "Z", c1)])
    return r1, r2


def parse(rec: bytes):
    ref_id, pos, l_name, mapq, bin_, n_cig, flag, l_seq, mref, mpos, tlen = struct.unpack_from("<iiBBHHHIiii", rec, 0)
    p = 32
    name = rec[p:p + l_name - 1].decode()
    p += l_name + 4 * n_cig
    seq = "".join("=ACMGRSVTWYHKDBN"[(rec[p + i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
    p += (l_seq + 1) // 2
    quals = list(rec[p:p + l_seq])
    p += l_seq
    tags = {}
    order = []
    while p + 3 <= len(rec):
        tag, ty = rec[p:p + 2].decode(), chr(rec[p + 2])
        p += 3
        if ty == "Z":
            e = rec.index(b"\0", p)
            val = rec[p:e].decode("latin1")
            p = e + 1
        elif ty in "cCsSiIf":
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}[ty]
            val = struct.unpack_from(fmt, rec, p)[0]
            p += struct.calcsize(fmt)
        elif ty == "B":
            sub = chr(rec[p])
            n = struct.unpack_from("<I", rec, p + 1)[0]
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            val = list(struct.unpack_from("<" + fmt * n, rec, p + 5))
            p += 5 + n * struct.calcsize(fmt)
        else:
            raise ValueError(ty)
        tags[tag] = (ty, val)
        order.append(tag)
    return dict(name=name, flag=flag, ref_id=ref_id, pos=pos, seq=seq, quals=quals, tags=tags, tag_order=order, bin=bin_, mapq=mapq,
                n_cigar=n_cig, mate_ref=mref, mate_pos=mpos, tlen=tlen)


def pair2(name, seq1, q1, seq2, q2, mi, start1, start2, rev1=False, rev2=True, ref_id=0, rx=None, extra=(), cigar1=None, cigar2=None):
    """`SamBuilder::add_pair()` (fgumi-sam/src/builder.rs:1642-1762): 1-based starts, sequences stored as given,
    MC tags, TLEN from the outer coordinates; tags in the order attr(s), MC."""
    L1, L2 = len(seq1), len(seq2)
    c1, c2 = cigar1 or f"{L1}M", cigar2 or f"{L2}M"
    q1 = [q1] * L1 if isinstance(q1, int) else q1
    q2 = [q2] * L2 if isinstance(q2, int) else q2

    def reflen(c):
        return sum(o >> 4 for o in cigar_ops(c) if (o & 15) in (0, 2, 3, 7, 8))

    p1, p2 = start1, start2
    e1, e2 = p1 + reflen(c1) - 1, p2 + reflen(c2) - 1
    left, right = (p1, e2) if p1 <= p2 else (p2, e1)
    tlen = right - left + 1
    t1 = tlen if p1 <= p2 else -tlen
    t2 = tlen if p2 <= p1 else -tlen
    tags = [("MI", "Z", mi)] + ([("RX", "Z", rx)] if rx else []) + list(extra)
    f1 = 0x1 | 0x40 | (0x10 if rev1 else 0) | (0x20 if rev2 else 0)
    f2 = 0x1 | 0x80 | (0x10 if rev2 else 0) | (0x20 if rev1 else 0)
    r1 = make_record(name, seq1, q1, flag=f1, ref_id=ref_id, pos=p1 - 1, cigar=c1, mate_ref=ref_id, mate_pos=p2 - 1, tlen=t1, tags=tags + [("MC", "Z", c2)])
    r2 = make_record(name, seq2, q2, flag=f2, ref_id=ref_id, pos=p2 - 1, cigar=c2, mate_ref=ref_id, mate_pos=p1 - 1, tlen=t2, tags=tags + [("MC", "Z", c1)])
    return r1, r2
