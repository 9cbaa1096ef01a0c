"""Field-level diff of the HIP simplex pipeline against the ORACLE on simulated families (debugging aid for kernel work; the
oracle is the checker, never the product).  usage: python tools/split_debug.py [quick]
Prints, per workload, OK or the first differing consensus record: family, record type, field (name / seq / qual / tag), and the
first differing positions; then the counters that differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamutil  # noqa: E402
import fgx_opts  # noqa: E402
import orc  # noqa: E402
from fgumi_amd import ConsensusCallingStats, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads, split_records  # noqa: E402


def fields(rec):
    p = bamutil.parse(rec)
    return p


def diff_records(a, b):
    pa, pb = bamutil.parse(a), bamutil.parse(b)
    out = []
    for k in sorted(set(pa) | set(pb)):
        va, vb = pa.get(k), pb.get(k)
        if va != vb:
            if isinstance(va, (str, bytes, list, tuple)) and isinstance(vb, type(va)) and len(va) == len(vb):
                pos = [i for i in range(len(va)) if va[i] != vb[i]]
                out.append(f"    {k}: {len(pos)} positions differ, first {pos[:8]}: got {[va[i] for i in pos[:8]]} want {[vb[i] for i in pos[:8]]}")
            else:
                sa, sb = repr(va), repr(vb)
                out.append(f"    {k}: got {sa[:200]} want {sb[:200]}")
    return out


def run(label, nf, opts_kw=None, caller_kw=None, **sim):
    g = simulate_grouped_reads(nf, **sim)
    kw = dict(min_reads=1, min_consensus_base_quality=2, cell_tag="CB")
    kw.update(caller_kw or {})
    caller = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(**kw), overlapping_consensus=True, device=0)
    out = caller.process_batch(g)
    st = caller.last_batch_statistics() if hasattr(caller, "last_batch_statistics") else None
    caller.close()
    o = fgx_opts.defaults(**dict(dict(min_reads=1), **(opts_kw or {})))
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first)
    ws = ConsensusCallingStats.from_array(want["stats"])
    stats_ok = st is None or (st.total_reads, st.consensus_reads, st.filtered_reads, st.rejection_reasons, st.overlapping) == \
        (ws.total_reads, ws.consensus_reads, ws.filtered_reads, ws.rejection_reasons, ws.overlapping)
    if not stats_ok:
        print(f"{label}: STATS differ\n   got  {st}\n   want {ws}")
    if out.data == want["data"] and out.count == want["count"]:
        print(f"{label}: {'OK' if stats_ok else 'records OK'} ({out.count} records, {len(out.data)} bytes)")
        return stats_ok
    got_recs, want_recs = split_records(out.data), split_records(want["data"])
    print(f"{label}: MISMATCH got {len(got_recs)} records / {len(out.data)} bytes, want {len(want_recs)} / {len(want['data'])}")
    shown = 0
    for i, (a, b) in enumerate(zip(got_recs, want_recs)):
        if a != b:
            print(f"  record {i}:")
            for line in diff_records(a, b)[:12]:
                print(line)
            shown += 1
            if shown >= 3:
                break
    nbad = sum(1 for a, b in zip(got_recs, want_recs) if a != b)
    print(f"  {nbad} of {min(len(got_recs), len(want_recs))} aligned records differ")
    return False


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = True
    ok &= run("depth8", 2000 if quick else 20000, family_size=8)
    ok &= run("depth8 err1%", 2000, family_size=8, error_rate_ppm=10000)
    ok &= run("depth5 insert200", 2000, family_size=5, insert_mean=200, insert_sd=40)
    ok &= run("depth20", 500, family_size=20)
    ok &= run("longtail 2..50", 1500, family_size=2, family_size_max=50)
    ok &= run("depth8 L100", 1000, family_size=8, read_length=100, insert_mean=180, insert_sd=30)
    ok &= run("depth8 L151", 1000, family_size=8, read_length=151)
    ok &= run("depth8 min_reads3", 1000, opts_kw=dict(min_reads=3), caller_kw=dict(min_reads=3), family_size=2, family_size_max=8)
    ok &= run("depth8 minbq0", 500, opts_kw=dict(min_input_base_quality=0), caller_kw=dict(min_input_base_quality=0), family_size=8)
    print("ALL OK" if ok else "SOME FAILED")
    sys.exit(0 if ok else 1)
