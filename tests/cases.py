"""Crafted MI groups covering the edge cases the reference tests."""
import random

import bamutil


def _rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def crafted_groups():
    rng = random.Random(7)
    groups = []
    tmpl = _rand_seq(rng, 400)
    # 1. fragments only (GATTACA-style, vanilla_caller.rs:3070-3139)
    groups.append([bamutil.frag("r1", "GATTACA", 10, "UMI1"), bamutil.frag("r2", "GATTACA", 10, "UMI1"), bamutil.frag("r3", "GATTTCA", 10, "UMI1")])
    # 2. differing lengths → shortened consensus with min_reads=2
    groups.append([bamutil.frag("a", "GATTACAGG", 30, "U2"), bamutil.frag("b", "GATTACA", 30, "U2")])
    # 3. indel CIGARs: minority alignment filtered
    groups.append([bamutil.frag("a", tmpl[:50], 35, "U3", cigar="50M"), bamutil.frag("b", tmpl[:50], 35, "U3", cigar="50M"),
                   bamutil.frag("c", tmpl[:50], 35, "U3", cigar="20M2I28M"), bamutil.frag("d", tmpl[:40], 35, "U3", cigar="40M"),
                   bamutil.frag("e", tmpl[:50], 35, "U3", cigar="5S45M")])
    # 4. tie in group sizes → smaller CIGAR wins
    groups.append([bamutil.frag("a", tmpl[:30], 35, "U4", cigar="10M1D20M"), bamutil.frag("b", tmpl[:30], 35, "U4", cigar="10M2D20M")])
    # 5. unmapped + mapped mix, secondary/supplementary records
    groups.append([bamutil.frag("a", tmpl[:30], 35, "U5"), bamutil.frag("b", tmpl[:30], 35, "U5", flag=0x4, cigar=""),
                   bamutil.frag("c", tmpl[:30], 35, "U5", flag=0x100), bamutil.frag("d", tmpl[:30], 35, "U5", flag=0x800)])
    # 6. reverse-strand fragments with IUPAC codes and N, low qualities (masking + trailing-N strip)
    groups.append([bamutil.frag("a", "ACGTRYNNACGTAC", [30] * 10 + [5, 5, 30, 5], "U6", flag=0x10),
                   bamutil.frag("b", "ACGTACGTACGTAC", [30] * 14, "U6", flag=0x10)])
    # 7. all-low-quality read → zero length after trimming
    groups.append([bamutil.frag("a", "ACGTACGT", 2, "U7"), bamutil.frag("b", "ACGTACGT", 30, "U7")])
    # 8. overlapping pair with agreement/disagreement + soft clips + insertion (overlap iterator paths)
    s1, s2 = tmpl[:100], tmpl[60:160]
    s2m = s2[:10] + ("A" if s2[10] != "A" else "C") + s2[11:]
    groups.append(list(bamutil.pair("p1", s1, 30, s2m, 25, "U8", pos1=1000, pos2=1060, rx="ACGT-TTTT")) +
                  list(bamutil.pair("p2", s1, 31, s2, 31, "U8", pos1=1000, pos2=1060, rx="ACGT-TTTA")))
    groups.append(list(bamutil.pair("q1", tmpl[:80], 30, tmpl[45:125], 30, "U9", pos1=2000, pos2=2050, cigar1="5S40M2I33M", cigar2="30M3D40M10S")) +
                  list(bamutil.pair("q2", tmpl[:80], 33, tmpl[45:125], 28, "U9", pos1=2000, pos2=2050, cigar1="5S40M2I33M", cigar2="30M3D40M10S")))
    # 9. read-through (insert shorter than read): mate clip on both ends
    groups.append(list(bamutil.pair("t1", tmpl[:100], 30, tmpl[:100], 30, "U10", pos1=3000, pos2=2990)) +
                  list(bamutil.pair("t2", tmpl[:100], 30, tmpl[:100], 30, "U10", pos1=3000, pos2=2990)))
    # 10. orphan: R1 ok, R2 insufficient (min_reads=2)
    r1a, r2a = bamutil.pair("o1", tmpl[:60], 30, tmpl[100:160], 30, "U11", pos1=4000, pos2=4100)
    r1b, _ = bamutil.pair("o2", tmpl[:60], 30, tmpl[100:160], 30, "U11", pos1=4000, pos2=4100)
    groups.append([r1a, r2a, r1b])
    # 11. cell barcode + long MI, quality > 93, group below min_reads
    groups.append([bamutil.frag("a", "ACGTACGT", 95, "12345/A", tags=[("CB", "Z", "CELL-1"), ("RX", "Z", "AAAA")]),
                   bamutil.frag("b", "ACGTACGT", 95, "12345/A", tags=[("CB", "Z", "CELL-2"), ("RX", "Z", "AAAC")])])
    groups.append([bamutil.frag("solo", "ACGTACGTAC", 35, "U13")])
    # 12. zero-length record and odd-length sequences
    groups.append([bamutil.frag("a", "", [], "U14"), bamutil.frag("b", "ACGTACG", 30, "U14"), bamutil.frag("c", "ACGTACG", 30, "U14")])
    return groups
