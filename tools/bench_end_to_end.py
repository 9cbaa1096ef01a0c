#!/usr/bin/env python3
"""BAM in → BAM out through the engine (BASELINE.md §3 timing 2): BGZF inflate + record-boundary walk on the host cores,
upload, consensus on the MI355X, download, BGZF level-1 deflate.  Every stage is timed; the slowest one is named.
usage: python tools/bench_end_to_end.py [--families 1000000] [--depth 8] [--threads N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=1000000)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--dir", default="/tmp/fgx_e2e")
    a = ap.parse_args()
    import numpy as np
    from fgumi_amd import GroupedReads, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, bgzf, simulate_grouped_reads
    os.makedirs(a.dir, exist_ok=True)
    T = a.threads or os.cpu_count() or 1
    refs = [(f"chr{i + 1}", 2147483647) for i in range(24)]
    g = simulate_grouped_reads(a.families, family_size=a.depth)
    gin, gout = os.path.join(a.dir, "grouped.bam"), os.path.join(a.dir, "consensus.bam")
    in_bytes = bgzf.write_bam(gin, bgzf.grouped_input_header(refs), refs, g.blob, threads=T)
    n_rec, raw_bytes = int(g.n_rec), int(g.blob.size)
    grp_first = g.grp_first.copy()          # MI grouping itself is fgx_group_records (4.4 G records/s on the device, tools/bench_grouping.py)
    del g
    caller = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    st = {}
    t0 = time.perf_counter()
    with open(gin, "rb") as f:
        raw = f.read()
    st["read_file"] = time.perf_counter() - t0
    t = time.perf_counter()
    nat = bgzf.native_inflate(raw, T)             # library entry: block-parallel zlib inflate, CRC32 / ISIZE checked
    data = memoryview(nat[0]) if nat is not None else bgzf.bgzf_decompress(raw, T)
    st["bgzf_inflate"] = time.perf_counter() - t
    t = time.perf_counter()
    import struct
    (l_text,) = struct.unpack_from("<i", data, 4)
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, p)
    p += 4
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, p)
        p += 8 + l_name
    rec_off, rec_len = bgzf.record_boundaries(data, p)
    st["record_boundaries"] = time.perf_counter() - t
    assert len(rec_off) == n_rec
    t = time.perf_counter()
    gr = GroupedReads(np.frombuffer(data, dtype=np.uint8), rec_off, rec_len, grp_first)
    out = caller.process_batch(gr)            # upload + kernels + download (C ABI host entry)
    st["engine_host_entry"] = time.perf_counter() - t
    tm = getattr(caller, "last_timing", None) or {}
    t = time.perf_counter()
    out_bytes = bgzf.write_bam(gout, bgzf.consensus_header("A", "Read group", 0, "fgumi simplex"), [], out.data, level=1, threads=T)
    st["bgzf_deflate_write"] = time.perf_counter() - t
    total = time.perf_counter() - t0
    slow = max(st, key=st.get)
    print(json.dumps(dict(metric="BAM in -> BAM out, simplex consensus, raw reads/s end to end", value=n_rec / total, unit="raw reads/s",
                          families=a.families, depth=a.depth, raw_reads=n_rec, host_threads=T, total_s=total, stages_s=st, bottleneck=slow,
                          input_bam_bytes=in_bytes, input_uncompressed_bytes=raw_bytes, output_bam_bytes=out_bytes, consensus_records=int(out.count),
                          engine_timing_ms=tm,
                          note="host side: fgx_bgzf_inflate / fgx_bgzf_deflate (block-parallel zlib, no libdeflate in the image) and the native block_size chain walk; "
                               "the device-resident consensus step of the same batch is bench.py's number")))
    caller.close()


if __name__ == "__main__":
    main()
