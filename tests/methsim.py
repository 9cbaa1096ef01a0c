"""Seeded EM-Seq / TAPs-like families against a random genome, for the methylation-mode parity tests: fragments, FR pairs and duplex
molecules aligned to several contigs, with C→T conversions on the reads of one orientation and G→A on the other, sequencing errors,
indel CIGARs, soft clips, read-through inserts (overlapping mates) and reads that run off the end of a contig."""
import random

import bamutil

F_PAIRED, F_REVERSE, F_MATE_REVERSE, F_FIRST, F_LAST = 0x1, 0x10, 0x20, 0x40, 0x80


def genome(rng, n_contigs=3, length=4000):
    out = []
    for _ in range(n_contigs):
        s = "".join(rng.choice("ACGT") for _ in range(length))
        # soft-masked and unknown stretches, as a FASTA has them
        i = rng.randrange(length - 300)
        s = s[:i] + s[i:i + 120].lower() + "N" * 30 + s[i + 150:]
        out.append(s.encode())
    return out


def _convert(rng, seq, c_from, c_to, rate):
    return "".join(c_to if (b == c_from and rng.random() < rate) else b for b in seq)


def _errors(rng, seq, rate):
    return "".join(rng.choice("ACGT") if rng.random() < rate else b for b in seq)


def _read_from(rng, contig, pos, length, cigar_kind, top_like, conv_rate, err_rate):
    """Stored (reference-orientation) bases of a read aligned at 0-based `pos` with a CIGAR of the requested kind."""
    ref = contig.decode().upper()

    def span(a, n):
        s = ref[max(0, a):max(0, a) + n] if a + n > 0 else ""
        return (s + "A" * n)[:n].replace("N", "A")

    if length < 24:
        cigar_kind = "M"
    if cigar_kind == "M":
        cigar, seq = f"{length}M", span(pos, length)
    elif cigar_kind == "D":
        a = rng.randint(5, length - 5)
        d = rng.randint(1, 4)
        cigar, seq = f"{a}M{d}D{length - a}M", span(pos, a) + span(pos + a + d, length - a)
    elif cigar_kind == "I":
        a = rng.randint(5, length - 8)
        ins = rng.randint(1, 3)
        cigar, seq = f"{a}M{ins}I{length - a - ins}M", span(pos, a) + "".join(rng.choice("ACGT") for _ in range(ins)) + span(pos + a, length - a - ins)
    else:   # soft clips at both ends
        s1, s2 = rng.randint(1, 6), rng.randint(0, 5)
        m = length - s1 - s2
        cigar = f"{s1}S{m}M" + (f"{s2}S" if s2 else "")
        seq = "".join(rng.choice("ACGT") for _ in range(s1)) + span(pos, m) + "".join(rng.choice("ACGT") for _ in range(s2))
    seq = _convert(rng, seq, "C", "T", conv_rate) if top_like else _convert(rng, seq, "G", "A", conv_rate)
    return _errors(rng, seq, err_rate), cigar


def simplex_groups(rng, contigs, n_groups, depth=(1, 7), read_len=(20, 90)):
    groups = []
    for g in range(n_groups):
        ref_id = rng.randrange(len(contigs) + (1 if rng.random() < 0.03 else 0))      # now and then a contig outside the header
        contig = contigs[min(ref_id, len(contigs) - 1)]
        L = rng.randint(*read_len)
        pos = rng.randint(0, len(contig) - 10) if rng.random() < 0.9 else len(contig) - rng.randint(1, L)   # may run off the end
        kind = rng.choice(["M"] * 6 + ["D", "I", "S"])
        n = rng.randint(*depth)
        conv = rng.choice([0.0, 0.3, 0.9, 1.0])
        mi = f"{g}"
        reads = []
        layout = rng.choice(["frag", "frag_rev", "pair", "pair_overlap"])
        for i in range(n):
            k = kind if rng.random() < 0.85 else "M"                                  # a minority alignment now and then
            l_i = L if rng.random() < 0.7 else rng.randint(max(12, L - 15), L)
            if layout in ("frag", "frag_rev"):
                seq, cigar = _read_from(rng, contig, pos, l_i, k, layout == "frag", conv, 0.01)
                q = [rng.choice([8, 20, 30, 37]) for _ in seq]
                reads.append(bamutil.make_record(f"f{g}_{i}", seq, q, flag=F_REVERSE if layout == "frag_rev" else 0, ref_id=ref_id, pos=pos, cigar=cigar,
                                                 tags=[("MI", "Z", mi), ("RX", "Z", "ACGT")]))
            else:
                gap = rng.randint(-l_i // 2, 60) if layout == "pair_overlap" else rng.randint(20, 120)
                pos2 = max(0, pos + l_i + gap - l_i) if layout == "pair_overlap" else pos + l_i + gap
                s1, c1 = _read_from(rng, contig, pos, l_i, k, True, conv, 0.01)
                s2, c2 = _read_from(rng, contig, pos2, l_i, "M", True, conv, 0.01)
                r1, r2 = bamutil.pair(f"p{g}_{i}", s1, [rng.choice([20, 30, 37]) for _ in s1], s2, [rng.choice([20, 30, 37]) for _ in s2], mi,
                                      pos1=pos, pos2=pos2, cigar1=c1, cigar2=c2, rx="AAC-GGT", ref_id=ref_id)
                reads += [r1, r2]
        groups.append(reads)
    return groups


def duplex_groups(rng, contigs, n_groups, depth=(0, 4), read_len=(25, 80)):
    groups = []
    for g in range(n_groups):
        ref_id = rng.randrange(len(contigs))
        contig = contigs[ref_id]
        L = rng.randint(*read_len)
        p1 = rng.randint(0, len(contig) - 2 * L - 150)
        p2 = p1 + L + rng.randint(-L // 3, 100)
        conv = rng.choice([0.0, 0.5, 1.0])
        na, nb = rng.randint(*depth), rng.randint(*depth)
        if na + nb == 0:
            na = 1
        kind = rng.choice(["M"] * 5 + ["D", "S"])
        reads = []

        def rec(name, seq, cigar, flag, pos, mpos, mi):
            return bamutil.make_record(name, seq, [rng.choice([25, 30, 37]) for _ in seq], flag=flag, ref_id=ref_id, pos=pos, mapq=60, cigar=cigar, mate_ref=ref_id,
                                       mate_pos=mpos, tags=[("MI", "Z", mi), ("RX", "Z", "ACG-TTA" if mi.endswith("A") else "TTA-ACG"), ("MC", "Z", f"{L}M")])
        for i in range(na):   # A strand: R1 forward at p1 (top), R2 reverse at p2 (top): C→T
            s1, c1 = _read_from(rng, contig, p1, L, kind, True, conv, 0.005)
            s2, c2 = _read_from(rng, contig, p2, L, "M", True, conv, 0.005)
            reads += [rec(f"a{g}_{i}", s1, c1, F_PAIRED | F_FIRST | F_MATE_REVERSE, p1, p2, f"{g}/A"), rec(f"a{g}_{i}", s2, c2, F_PAIRED | F_LAST | F_REVERSE, p2, p1, f"{g}/A")]
        for i in range(nb):   # B strand: R1 reverse at p2 (bottom), R2 forward at p1 (bottom): G→A
            s1, c1 = _read_from(rng, contig, p2, L, "M", False, conv, 0.005)
            s2, c2 = _read_from(rng, contig, p1, L, kind, False, conv, 0.005)
            reads += [rec(f"b{g}_{i}", s1, c1, F_PAIRED | F_FIRST | F_REVERSE, p2, p1, f"{g}/B"), rec(f"b{g}_{i}", s2, c2, F_PAIRED | F_LAST | F_MATE_REVERSE, p1, p2, f"{g}/B")]
        groups.append(reads)
    return groups


def seeded(seed):
    return random.Random(seed)
