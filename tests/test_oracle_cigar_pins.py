"""Oracle pins: the unit tests of `clip_cigar_ops_raw` and `read_pos_at_ref_pos_raw` (`crates/fgumi-raw-bam/src/cigar.rs`,
`mod tests`) — the virtual hard clip and the reference-to-read coordinate map the CODEC caller lays its two strands out with
(`codec_caller.rs:625-1004`); the device CODEC kernel uses their closed forms for single-M reads, the general path the op walk."""
import numpy as np
import pytest

import bamutil
import orc


def ops(c):
    return np.array(bamutil.cigar_ops(c), dtype=np.uint32) if c else np.zeros(0, dtype=np.uint32)


def text(a):
    return "".join(f"{int(o) >> 4}{'MIDNSHP=X'[int(o) & 15]}" for o in a)


def clip(cigar, amount, from_start):
    a = ops(cigar)
    out = np.zeros(16, dtype=np.uint32)
    rc = np.zeros(1, dtype=np.uint64)
    n = orc.lib.orc_clip_cigar_ops(orc.ptr(a) if len(a) else None, len(a), amount, int(from_start), orc.ptr(out), 16, orc.ptr(rc))
    return text(out[:n]), int(rc[0])


@pytest.mark.parametrize("cigar,amount,from_start,want,ref_consumed", [
    ("10M", 0, True, "10M", 0),                       # test_clip_cigar_ops_raw_zero_clip
    ("", 5, True, "", 0),                             # ..._empty_cigar
    ("5S10M", 3, True, "3H2S10M", 0),                 # ..._upgrade_path_from_start
    ("10M5S", 3, False, "10M2S3H", 0),                # ..._upgrade_path_from_end
    ("10M", 3, True, "3H7M", 3),                      # ..._alignment_clip_from_start
    ("10M", 3, False, "7M3H", 0),                     # ..._alignment_clip_from_end
    ("2S10M", 5, True, "5H7M", 3),                    # ..._clip_past_existing_from_start
    ("10M2S", 5, False, "7M5H", 0),                   # ..._clip_past_existing_from_end
    ("10M3I5M", 10, True, "10H3I5M", 10),             # ..._with_insertion_at_boundary_start
    ("5M2D10M", 5, True, "5H10M", 7),                 # ..._with_deletion_at_boundary_start
    ("10M2D5M", 5, False, "10M5H", 0),                # ..._with_deletion_at_boundary_end
    ("10M", 4, True, "4H6M", 4),                      # ..._split_match_from_start
    ("10M", 4, False, "6M4H", 0),                     # ..._split_match_from_end
    ("5M3I5M", 6, True, "8H5M", 5),                   # ..._insertion_consumed_at_boundary_start
    ("5=3X", 4, True, "4H1=3X", 4),                   # ..._with_eq_and_x_ops
    ("10M", 10, True, "10H", 10),                     # ..._clip_entire_alignment_from_start
    ("10M", 10, False, "10H", 0),                     # ..._clip_entire_alignment_from_end
    ("3S10M2I5M4S", 8, True, "8H5M2I5M4S", 5),        # ..._complex_cigar_start
    ("3S10M2I5M4S", 8, False, "3S10M2I1M8H", 0),      # ..._complex_cigar_end
])
def test_clip_cigar_ops_raw(cigar, amount, from_start, want, ref_consumed):  # cigar.rs `test_clip_cigar_ops_raw_*`
    assert clip(cigar, amount, from_start) == (want, ref_consumed)


@pytest.mark.parametrize("cigar,ref_pos,last_if_deleted,want", [
    ("10M", 100, False, 1), ("10M", 105, False, 6), ("10M", 109, False, 10),      # ..._simple_match
    ("10M", 99, False, None), ("10M", 110, False, None),                          # ..._before_alignment, ..._past_alignment
    ("5M3D5M", 106, False, None), ("5M3D5M", 106, True, 5),                       # ..._in_deletion
    ("5M3I5M", 104, False, 5), ("5M3I5M", 105, False, 9),                         # ..._with_insertion
    ("3S10M", 100, False, 4),                                                     # ..._with_soft_clip
    ("2D5M", 100, True, 1),                                                       # ..._deletion_at_start
])
def test_read_pos_at_ref_pos_raw(cigar, ref_pos, last_if_deleted, want):  # cigar.rs `test_read_pos_at_ref_pos_raw_*` (alignment start 100)
    a = ops(cigar)
    got = orc.lib.orc_read_pos_at_ref_pos(orc.ptr(a), len(a), 100, ref_pos, int(last_if_deleted))
    assert (got or None) == want
