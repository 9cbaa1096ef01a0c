"""Differential fuzz of the general path's host orchestration (tests/hostemu: fgumi_amd/csrc/{simplex,duplex,codec}_host.cpp with the
kernels' work done on the host) against the oracle over hostile MI groups: every flag combination, random CIGARs over all nine ops,
missing / foreign / garbage MC tags, secondary and supplementary records, unmapped reads and mates, reads on other contigs, IUPAC codes
and no-calls, low and absent qualities, empty reads, RX values with and without dashes — under random option sets, with and without
the methylation-aware mode (reference = the template the reads were drawn from).  Where the
reference raises (the oracle throws), the product path must refuse the batch too.  FGX_FUZZ_ROUNDS scales the run (default: a few
seconds)."""
import os
import random

import numpy as np
import pytest

import bamutil
import fgx_opts
import hostemu
import orc
from fgumi_amd import GroupedReads
from fgumi_amd.caller import split_records

ROUNDS = int(os.environ.get("FGX_FUZZ_ROUNDS", "12"))
TMPL = "".join(random.Random(3).choice("ACGT") for _ in range(3000))


def random_cigar(rng, qlen, exotic):
    """A CIGAR whose query-consuming ops sum to qlen."""
    if qlen == 0:
        return ""
    if not exotic or qlen < 8:
        return f"{qlen}M"
    ops, left = [], qlen
    if rng.random() < 0.2:
        ops.append(f"{rng.randint(1, 4)}H")
    if rng.random() < 0.35 and left > 6:
        n = rng.randint(1, min(6, left - 4))
        ops.append(f"{n}S")
        left -= n
    tail = 0
    if rng.random() < 0.3 and left > 6:
        tail = rng.randint(1, min(6, left - 4))
        left -= tail
    while left > 0:
        n = rng.randint(1, left)
        ops.append(f"{n}{rng.choice('MMMM=X')}")
        left -= n
        if left > 0:
            k = rng.random()
            if k < 0.3:
                ops.append(f"{rng.randint(1, 5)}D")
            elif k < 0.4:
                ops.append(f"{rng.randint(1, 30)}N")
            elif k < 0.45:
                ops.append(f"{rng.randint(1, 3)}P")
            elif k < 0.75 and left > 1:
                i = rng.randint(1, min(4, left - 1))
                ops.append(f"{i}I")
                left -= i
    if tail:
        ops.append(f"{tail}S")
    if rng.random() < 0.1:
        ops.append(f"{rng.randint(1, 3)}H")
    return "".join(ops)


def random_read(rng, name, mi, kind, pos, L, exotic, paired, first, reverse, mate_pos, mate_cigar, mate_reverse):
    seq = "".join(rng.choice("ACGTNRYacgt") if rng.random() < 0.03 else TMPL[(pos + i) % len(TMPL)] for i in range(L)) if pos >= 0 else "".join(rng.choice("ACGT") for _ in range(L))
    if rng.random() < 0.05:
        seq = "".join(rng.choice("ACGT") if rng.random() < 0.3 else b for b in seq)
    quals = [rng.choice([2, 5, 9, 10, 11, 20, 30, 37, 41, 60, 93]) if rng.random() < 0.3 else rng.randint(25, 40) for _ in range(L)]
    if rng.random() < 0.02:
        quals = [2] * L                                           # zero length after masking
    flag = 0
    if paired:
        flag |= 0x1 | (0x40 if first else 0x80) | (0x20 if mate_reverse else 0)
        if rng.random() < 0.9:
            flag |= 0x2
    if reverse:
        flag |= 0x10
    r = rng.random()
    unmapped = exotic and r < 0.04
    if unmapped:
        flag |= 0x4
    if exotic and 0.04 <= r < 0.07:
        flag |= 0x100
    if exotic and 0.07 <= r < 0.10:
        flag |= 0x800
    if exotic and paired and rng.random() < 0.04:
        flag |= 0x8
    if exotic and rng.random() < 0.02:
        flag |= 0x200 | 0x400
    cigar = "" if unmapped else random_cigar(rng, L, exotic and rng.random() < 0.5)
    tags = [("MI", "Z", mi)]
    if rng.random() < 0.8:
        rx = rng.choice(["ACGT-TTGA", "ACGT-TTGA", "ACGA-TTGA", "NNGT-TTGA", "acgt-TTGA"])
        if rng.random() < 0.0001:
            rx = rng.choice(["ACGTAC", "ACGT+TTGA"])          # unequal lengths / a foreign separator: the reference panics, the product refuses
        tags.append(("RX", "Z", rx))
    if rng.random() < 0.5:
        tags.append(("CB", "Z", rng.choice(["CELL1", "CELL2"])))
    if paired and rng.random() < 0.85:
        mc = mate_cigar if rng.random() < 0.9 else rng.choice(["*", "10M5", "abc", "", "300M", "5S20M"])
        tags.append(("MC", "Z", mc))
    if rng.random() < 0.3:
        tags.append(("NM", "i", rng.randint(0, 300)))
    ref_id = 0 if (not exotic or rng.random() < 0.95) else rng.choice([1, -1])
    mref = ref_id if rng.random() < 0.97 else 1
    tlen = (mate_pos - pos) if paired else 0
    return bamutil.make_record(name, seq, quals, flag=flag, ref_id=ref_id, pos=pos, mapq=rng.choice([0, 30, 60]), cigar=cigar, mate_ref=mref if paired else -1,
                               mate_pos=mate_pos if paired else -1, tlen=tlen if not reverse else -abs(tlen), tags=tags)


def random_group(rng, g, kind, exotic):
    mi_base = f"{g}"
    recs = []
    n_templates = rng.choice([1, 1, 2, 3, 4, 6, 9]) if rng.random() < 0.9 else 0
    base_pos = rng.randint(0, 2000) if rng.random() < 0.97 else -1
    L = rng.choice([0, 1, 7, 20, 33, 64, 65, 100, 151]) if exotic and rng.random() < 0.2 else rng.randint(20, 120)
    insert = rng.choice([L, L + 10, 2 * L, 2 * L + 50, max(5, L // 2)])
    for t in range(n_templates):
        strand = rng.choice("AB")
        mi = mi_base if kind == "simplex" else (f"{mi_base}/{strand}" if kind == "duplex" else mi_base)
        if kind == "duplex" and exotic and rng.random() < 0.0004:
            mi = mi_base                                      # no strand suffix: the reference refuses the batch
        layout = rng.random()
        name = f"t{g}_{t}"
        Lt = L if rng.random() < 0.75 else max(0, L - rng.randint(0, 12))
        if layout < 0.15 and kind != "codec":
            recs.append(random_read(rng, name, mi, kind, base_pos, Lt, exotic, False, True, rng.random() < 0.3, -1, "", False))
            continue
        p1 = base_pos + (rng.randint(-3, 3) if exotic and rng.random() < 0.2 else 0)
        p2 = max(0, p1 + insert - Lt) if p1 >= 0 else -1
        c2 = f"{Lt}M"
        swap = kind == "duplex" and strand == "B"              # B strand: R1 is the reverse read at the far end
        rev1 = swap if rng.random() < 0.95 else not swap
        r1 = random_read(rng, name, mi, kind, p2 if swap else p1, Lt, exotic, True, True, rev1, p1 if swap else p2, c2, not rev1)
        r2 = random_read(rng, name, mi, kind, p1 if swap else p2, Lt, exotic, True, False, not rev1, p2 if swap else p1, c2, rev1)
        if rng.random() < 0.93:
            recs += [r1, r2]
        else:
            recs.append(rng.choice([r1, r2]))                  # a mate is missing
    rng.shuffle(recs) if (exotic and rng.random() < 0.3) else None
    return recs


def random_options(rng, kind):
    kw = dict(min_input_base_quality=rng.choice([0, 10, 10, 20, 30]), produce_per_base_tags=rng.randint(0, 1), trim=int(rng.random() < 0.25),
              overlapping_consensus=rng.randint(0, 1), track_rejects=rng.randint(0, 1), cell_tag=rng.choice([b"CB", b"\0\0"]),
              error_rate_pre_umi=rng.choice([45, 45, 30, 93]), error_rate_post_umi=rng.choice([40, 40, 20, 60]), tie_rule=rng.randint(0, 1),
              read_name_prefix=rng.choice([b"", b"fz"]))
    if kind == "simplex":
        kw.update(kind=0, min_reads=rng.choice([1, 1, 2, 3]), max_reads=rng.choice([-1, -1, 2, 3, 5]), min_consensus_base_quality=rng.choice([0, 2, 2, 20, 40]))
        return fgx_opts.defaults(**kw)
    if kind == "duplex":
        kw.update(kind=1, duplex_max_reads_per_strand=rng.choice([-1, -1, 1, 2, 4]))
        o = fgx_opts.defaults(**kw)
        mr = rng.choice([(1, 1, 0), (1, 1, 0), (1, 1, 1), (2, 1, 1), (3, 2, 1), (2, 2, 0), (4, 2, 2)])
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
        return o
    kw.pop("overlapping_consensus")
    kw.pop("trim")
    kw.update(kind=2, codec_min_reads_per_strand=rng.choice([1, 1, 2]), codec_max_reads_per_strand=rng.choice([-1, -1, 2, 3]), codec_min_duplex_length=rng.choice([1, 1, 10, 30]),
              codec_outer_bases_length=rng.choice([0, 5, 5, 10]))
    if rng.random() < 0.3:
        kw.update(codec_has_single_strand_qual=1, codec_single_strand_qual=rng.choice([5, 10, 30]))
    if rng.random() < 0.3:
        kw.update(codec_has_outer_bases_qual=1, codec_outer_bases_qual=rng.choice([3, 7, 20]))
    if rng.random() < 0.3:
        kw.update(codec_max_duplex_disagreements=rng.choice([0, 1, 3, 10]))
    if rng.random() < 0.3:
        kw.update(codec_max_duplex_disagreement_rate=rng.choice([0.0, 0.01, 0.05, 0.5]))
    return fgx_opts.defaults(**kw)


def differential(kind, seed, n_groups=60):
    rng = random.Random(seed)
    exotic = rng.random() < 0.6
    groups = [random_group(rng, g, kind, exotic) for g in range(n_groups)]
    groups = [x for x in groups if x] or [[random_read(rng, "solo", "0/A" if kind == "duplex" else "0", kind, 10, 30, False, False, True, False, -1, "", False)]]
    o = random_options(rng, kind)
    contigs = None
    if kind != "codec" and rng.random() < 0.4:                 # the methylation-aware mode over the same hostile groups
        o.methylation_mode = rng.choice([1, 2])
        contigs = [TMPL.encode() if rng.random() < 0.8 else TMPL.lower().encode(), TMPL[500:900].encode()][:rng.choice([1, 2, 2])]
    g = GroupedReads.from_groups(groups)
    batch = {"simplex": 50, "duplex": 100, "codec": 1000}[kind]
    orc.set_reference(contigs)
    try:
        want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=max(batch, n_groups))
    except RuntimeError as e:
        with pytest.raises(RuntimeError):
            hostemu.process(o, contigs, g)
        return "error: " + str(e)[:60]
    finally:
        orc.set_reference(None)
    got = hostemu.process(o, contigs, g)
    if got["data"] != want["data"]:
        a, b = split_records(got["data"]), split_records(want["data"])
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                raise AssertionError(f"{kind} seed {seed}: record {i} differs:\n got {bamutil.parse(x)}\nwant {bamutil.parse(y)}")
        raise AssertionError(f"{kind} seed {seed}: record count differs {len(a)} vs {len(b)}")
    assert got["count"] == want["count"]
    assert np.array_equal(got["stats"], want["stats"]), (kind, seed, got["stats"].tolist(), want["stats"].tolist())
    if o.track_rejects:
        assert got["rejects"] == want["rejects"] and got["n_rejects"] == want["n_rejects"], (kind, seed)
    return want["count"]


@pytest.mark.parametrize("kind", ["simplex", "duplex", "codec"])
def test_general_path_equals_the_oracle_on_hostile_groups(kind):
    produced = 0
    for seed in range(ROUNDS):
        r = differential(kind, 1000 * (["simplex", "duplex", "codec"].index(kind) + 1) + seed)
        if isinstance(r, int):
            produced += r
    assert produced > 0
