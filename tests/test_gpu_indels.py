"""GPU parity: families whose reads carry indels / skips / clips stay on the device (workgroup-per-family kernel: general CIGAR
parse, mate clip, overlap correction through the CIGARs, alignment filter = select_most_common_alignment_group) and are
byte-identical to the oracle; nothing is deferred to the host path."""
import random

import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import GroupedReads, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, split_records

pytestmark = pytest.mark.gpu

TMPL = "".join(random.Random(11).choice("ACGT") for _ in range(4000))


def qlen(cigar):
    return sum(ln >> 4 for ln in bamutil.cigar_ops(cigar) if (ln & 15) in (0, 1, 4, 7, 8))


def rlen(cigar):
    return sum(ln >> 4 for ln in bamutil.cigar_ops(cigar) if (ln & 15) in (0, 2, 3, 7, 8))


def fr_pair(rng, name, mi, start, insert, c1, c2, q=(20, 41), extra=()):
    """FR pair on TMPL: R1 forward at `start`, R2 reverse ending at start+insert; SEQ in stored orientation."""
    l1, l2 = qlen(c1), qlen(c2)
    p2 = start + insert - rlen(c2)
    s1 = "".join(rng.choice("ACGT") if rng.random() < 0.02 else TMPL[(start + i) % 4000] for i in range(l1))
    s2 = "".join(rng.choice("ACGT") if rng.random() < 0.02 else TMPL[(p2 + i) % 4000] for i in range(l2))
    q1 = [rng.randint(*q) for _ in range(l1)]
    q2 = [rng.randint(*q) for _ in range(l2)]
    return list(bamutil.pair(name, s1, q1, s2, q2, mi, pos1=start, pos2=p2, cigar1=c1, cigar2=c2, rx="ACGTACGT", extra=extra))


def indel_groups(seed=5, n_groups=160, max_pairs=12):
    rng = random.Random(seed)
    C1 = ["100M", "40M2D60M", "40M2D60M", "40M3I57M", "5S95M", "5S35M2D60M", "30M1I29M1D40M", "3H100M", "50M10N50M", "40M2D58M2S", "100M", "39M3D61M"]
    C2 = ["100M", "60M2D40M", "60M2D40M", "57M3I40M", "95M5S", "60M2D35M5S", "100M", "100M2H", "50M10N50M", "100M", "2S40M1D58M", "100M"]
    groups = []
    for g in range(n_groups):
        npairs = rng.randint(1, max_pairs)
        start = rng.randint(10, 3000)
        insert = rng.choice([120, 150, 180, 260])
        major = rng.randrange(len(C1))
        recs = []
        for k in range(npairs):
            ci = major if rng.random() < 0.75 else rng.randrange(len(C1))      # mostly one alignment, some minority reads
            recs += fr_pair(rng, f"g{g}r{k}", str(g), start, insert, C1[ci], C2[ci])
        groups.append(recs)
    return groups


def _run(groups, **kw):
    g = GroupedReads.from_groups(groups)
    opt = VanillaUmiConsensusOptions(min_reads=kw.get("min_reads", 1), min_consensus_base_quality=2, cell_tag="CB", trim=kw.get("trim", False))
    c = VanillaUmiConsensusCaller("", "A", opt, overlapping_consensus=kw.get("overlapping", True))
    out = c.process_batch_device(g.to_device())
    data = out.to_host()
    st = c.last_batch_statistics()
    c.close()
    want = orc.process(fgx_opts.defaults(min_reads=kw.get("min_reads", 1), overlapping_consensus=int(kw.get("overlapping", True)), trim=int(kw.get("trim", False))),
                       g.blob, g.rec_off, g.rec_len, g.grp_first)
    assert out.n_deferred == kw.get("expect_deferred", 0), out.n_deferred
    if kw.get("expect_deferred", 0):
        return
    if data != want["data"]:
        a, b = split_records(data), split_records(want["data"])
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(x)}\nwant {bamutil.parse(y)}")
        raise AssertionError(f"record count differs {len(a)} vs {len(b)}")
    arr = np.zeros(28, dtype=np.uint64)
    arr[0], arr[1], arr[2] = st.total_reads, st.consensus_reads, st.filtered_reads
    for r, v in st.rejection_reasons.items():
        arr[3 + int(r)] = v
    arr[24:28] = [st.overlapping[k] for k in ("overlapping_bases", "bases_agreeing", "bases_disagreeing", "bases_corrected")]
    assert np.array_equal(arr, want["stats"]), (arr.tolist(), want["stats"].tolist())
    return st


@pytest.mark.parametrize("kw", [dict(), dict(min_reads=2), dict(min_reads=3, overlapping=False), dict(trim=True)])
def test_indel_families_stay_on_device(kw):
    st = _run(indel_groups(), **kw)
    assert st.rejection_reasons.get(6, 0) > 0          # MinorityAlignment: the filter dropped reads


def test_alignment_filter_ties_and_orientation():
    rng = random.Random(2)
    groups = []
    # two groups of equal size: the smaller CIGAR wins; three-way ties; reverse-strand reads reverse their CIGARs
    for g, cs in enumerate([["40M2D60M", "40M2D60M", "40M3D60M", "40M3D60M"], ["100M", "50M1I49M", "50M2I48M"], ["30M5D70M"] * 3 + ["100M"] * 3,
                            ["10S90M", "100M", "90M10S", "40M2D60M"], ["20M1D80M", "20M1D60M20S", "20M1D80M", "95M5S"]]):
        recs = []
        for k, c in enumerate(cs):
            recs += fr_pair(rng, f"t{g}r{k}", f"T{g}", 100 + g * 300, 160, c, c)
        groups.append(recs)
    # fragments (unpaired) of different lengths: shorter reads are prefixes of the longer ones' CIGARs
    fr = []
    for k, (L, c) in enumerate([(100, "100M"), (80, "80M"), (100, "60M2D40M"), (70, "60M2D10M"), (90, "60M2D30M"), (100, "60M2D40M")]):
        fr.append(bamutil.frag(f"f{k}", TMPL[500:500 + L], [30 + k] * L, "F", pos=500, cigar=c))
        fr.append(bamutil.frag(f"r{k}", TMPL[500:500 + L], [30 + k] * L, "F", pos=500, cigar=c, flag=0x10))
    groups.append(fr)
    _run(groups)
    _run(groups, min_reads=2)


def test_large_family_with_indels_and_clips():
    rng = random.Random(9)
    recs = []
    for k in range(50):          # 100 records: beyond one wavefront
        c = "40M2D60M" if k % 5 else "100M"
        recs += fr_pair(rng, f"L{k}", "big", 700, 170, c, "5S95M" if k % 7 == 0 else "100M")
    _run([recs, recs[:40], fr_pair(rng, "solo", "one", 50, 150, "20M4D80M", "100M")])


def test_long_cigars_stay_on_the_device():
    """Round 5: reads of 7 .. 16 CIGAR ops (and mates whose MC tag is as long) are the workgroup kernel's — round 4 sent every family that
    held one to the general path.  Families of one such pair, and families in which the long alignment is the majority / the minority of
    the alignment filter; both mates long; clips around the indels."""
    rng = random.Random(1)
    nine = "10M1D10M1D10M1D10M1D60M"                             # 9 ops
    fifteen = "5M1D" * 7 + "65M"                                # 15 ops
    sixteen = "2S" + "6M1I" * 7 + "49M"                         # 16 ops, 100 query bases
    assert qlen(nine) == 100 and qlen(fifteen) == 100 and qlen(sixteen) == 100
    groups = [fr_pair(rng, "x", "1", 100, 170, nine, "100M")]
    groups.append(fr_pair(rng, "y", "2", 300, 180, fifteen, nine))
    groups.append(fr_pair(rng, "z", "3", 500, 170, sixteen, sixteen))
    for g, (major, minor) in enumerate([(nine, "100M"), ("100M", fifteen), (sixteen, nine)]):
        recs = []
        for k in range(7):
            c = major if k % 3 else minor
            recs += fr_pair(rng, f"m{g}r{k}", f"M{g}", 800 + 200 * g, 175, c, "100M" if k % 2 else c)
        groups.append(recs)
    _run(groups)
    _run(groups, min_reads=2)
    _run(groups, overlapping=False)


def test_more_than_sixteen_cigar_ops_defer_to_the_host():
    rng = random.Random(1)
    seventeen = "5M1D" * 8 + "60M"
    assert qlen(seventeen) == 100 and len(bamutil.cigar_ops(seventeen)) == 17
    _run([fr_pair(rng, "x", "1", 100, 170, seventeen, "100M")], expect_deferred=1)


def random_cigar(rng, L):
    """A CIGAR of at most 6 ops that consumes exactly L query bases (clips, indels, skips and pads in legal and odd places)."""
    for _ in range(100):
        ops = []
        if rng.random() < 0.15:
            ops.append((rng.randint(1, 5), "H"))
        if rng.random() < 0.3:
            ops.append((rng.randint(1, 12), "S"))
        n_mid = rng.choice([1, 1, 1, 2, 2, 3])
        for k in range(n_mid):
            ops.append((0, "M"))
            if k + 1 < n_mid:
                ops.append((rng.randint(1, 6), rng.choice("IDDINP")))
        if rng.random() < 0.3:
            ops.append((rng.randint(1, 12), "S"))
        if rng.random() < 0.1:
            ops.append((rng.randint(1, 5), "H"))
        if len(ops) > 6:
            continue
        fixed = sum(n for n, t in ops if t in "SI")
        n_m = sum(1 for n, t in ops if t == "M")
        if L - fixed < n_m:
            continue
        rest = L - fixed
        cuts = sorted(rng.sample(range(1, rest), n_m - 1)) if n_m > 1 else []
        lens = [b - a for a, b in zip([0] + cuts, cuts + [rest])]
        it = iter(lens)
        return "".join(f"{next(it) if t == 'M' else n}{'=' if t == 'M' and rng.random() < 0.1 else t}" for n, t in ops)
    return f"{L}M"


def test_random_cigars_differential():
    rng = random.Random(77)
    groups = []
    for g in range(300):
        start = rng.randint(10, 3000)
        insert = rng.choice([90, 130, 170, 240])
        L = rng.choice([60, 100, 101])
        shared = [random_cigar(rng, L) for _ in range(2)]
        recs = []
        for k in range(rng.randint(1, 10)):
            c1 = shared[0] if rng.random() < 0.7 else random_cigar(rng, L)
            c2 = shared[1] if rng.random() < 0.7 else random_cigar(rng, L)
            pr = fr_pair(rng, f"z{g}r{k}", f"Z{g}", start + rng.choice([0, 0, 0, 1, 3]), insert, c1, c2, q=(2, 41))
            if rng.random() < 0.1:
                pr = pr[:1] if rng.random() < 0.5 else pr[1:]          # orphan mates
            recs += pr
        if rng.random() < 0.2:
            recs.append(bamutil.frag(f"frag{g}", TMPL[start:start + L], [33] * L, f"Z{g}", pos=start, cigar=random_cigar(rng, L)))
        groups.append(recs)
    for kw in (dict(), dict(min_reads=2), dict(overlapping=False, trim=True)):
        _run(groups, **kw)


@pytest.mark.parametrize("max_reads,min_reads", [(2, 1), (3, 2), (1, 1), (0, 1), (5, 6)])
def test_max_reads_downsampling_on_device(max_reads, min_reads):
    """--max-reads that bites: the fgbio name-rank selection (Murmur3_32 over UTF-16 units, lowest ranks stay, ties in file order) runs in
    the wave kernel and, for families with indels or more than 64 records, in the workgroup kernel; nothing is deferred."""
    from fgumi_amd import simulate_grouped_reads
    g0 = simulate_grouped_reads(400, family_size=1, family_size_max=9, error_rate_ppm=5000)
    groups = [g0.records(i) for i in range(g0.n_grp)] + indel_groups(seed=8, n_groups=60, max_pairs=10)
    rng = random.Random(12)
    big = []
    for k in range(45):
        big += fr_pair(rng, f"B{k}", "big", 900, 170, "100M", "100M")
    groups.append(big)
    g = GroupedReads.from_groups(groups)
    opt = VanillaUmiConsensusOptions(min_reads=min_reads, max_reads=max_reads, min_consensus_base_quality=2, cell_tag="CB")
    c = VanillaUmiConsensusCaller("", "A", opt, overlapping_consensus=True)
    out = c.process_batch_device(g.to_device())
    data = out.to_host()
    st = c.last_batch_statistics()
    c.close()
    want = orc.process(fgx_opts.defaults(min_reads=min_reads, max_reads=max_reads), g.blob, g.rec_off, g.rec_len, g.grp_first)
    assert out.n_deferred == 0
    assert out.count == want["count"] and data == want["data"]
    assert st.rejection_reasons.get(19, 0) == int(want["stats"][3 + 19]) and st.filtered_reads == int(want["stats"][2])
    if max_reads < 9:
        assert st.rejection_reasons.get(19, 0) > 0
