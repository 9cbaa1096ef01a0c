#!/usr/bin/env python3
"""Direct records (fastpath.h) against the column-scratch path and the oracle, on the GPU box, with a field-level diff of the first record
that differs (TEST / DEBUG TOOLING: it uses tests/orc.py, the oracle binding).

  python tools/direct_check.py [--families 3000]

Each case runs the same device-resident batch twice in one process — FGX_DIRECT=0 (round-3 chain: scratch + k_emit) and the default —
and compares both with the oracle's bytes; `last_direct` says which way the second run really went (1 direct, 2 direct + merge)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import bamutil  # noqa: E402
import fgx_opts  # noqa: E402
import orc  # noqa: E402
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, lib, simulate_grouped_reads, split_records  # noqa: E402

lib.fgx_debug_last_direct.restype = C.c_int
lib.fgx_debug_last_direct.argtypes = [C.c_void_p]


def first_diff(a: bytes, b: bytes):
    try:
        ra, rb = split_records(a), split_records(b)
    except Exception as e:                                   # (a stream whose block sizes do not chain)
        k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
        return f"streams differ from byte {k} ({len(a)} vs {len(b)} bytes); unparsable: {e}"
    for i, (x, y) in enumerate(zip(ra, rb)):
        if x != y:
            try:
                px, py = bamutil.parse(x), bamutil.parse(y)
            except Exception as e:
                k = next((j for j in range(min(len(x), len(y))) if x[j] != y[j]), min(len(x), len(y)))
                return f"record {i}: {len(x)} vs {len(y)} bytes, first differing byte {k}; unparsable ({e})\n   got  {x[:96].hex()}\n   want {y[:96].hex()}"
            keys = [k for k in px if k != "tags" and px[k] != py[k]]
            tags = [t for t in set(px["tags"]) | set(py["tags"]) if px["tags"].get(t) != py["tags"].get(t)]
            msg = f"record {i} of {len(ra)} / {len(rb)} differs in fields {keys} and tags {tags}"
            for k in keys:
                if k in ("seq", "quals"):
                    pos = [j for j in range(min(len(px[k]), len(py[k]))) if px[k][j] != py[k][j]]
                    msg += f"\n   {k}: lengths {len(px[k])} / {len(py[k])}, differing positions {pos[:20]}"
                    for j in pos[:4]:
                        msg += f"\n      [{j}] got {px[k][j]!r} want {py[k][j]!r}"
                else:
                    msg += f"\n   {k}: got {px[k]!r} want {py[k]!r}"
            for t in tags:
                gx, gy = px["tags"].get(t), py["tags"].get(t)
                if gx and gy and isinstance(gx[1], list) and isinstance(gy[1], list):
                    pos = [j for j in range(min(len(gx[1]), len(gy[1]))) if gx[1][j] != gy[1][j]]
                    msg += f"\n   {t}: lengths {len(gx[1])} / {len(gy[1])}, differing positions {pos[:20]} got {[gx[1][j] for j in pos[:6]]} want {[gy[1][j] for j in pos[:6]]}"
                else:
                    msg += f"\n   {t}: got {gx!r} want {gy!r}"
            return msg
    return f"{len(ra)} vs {len(rb)} records, common prefix equal"


def run(name, g, vo_kw, oracle_kw):
    want = orc.process(fgx_opts.defaults(**oracle_kw), g.blob, g.rec_off, g.rec_len, g.grp_first)
    res = {}
    for mode in ("0", "1"):
        os.environ["FGX_DIRECT"] = mode
        c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(**vo_kw), overlapping_consensus=oracle_kw.get("overlapping_consensus", 1) != 0)
        out = c.process_batch_device(g.to_device())
        res[mode] = (out.to_host(), int(out.n_deferred), np.array(c.last_stats_array, dtype=np.uint64), lib.fgx_debug_last_direct(c._h))
        c.close()
    ok = True
    for mode in ("0", "1"):
        data, ndef, st, how = res[mode]
        same = data == want["data"] and ndef == 0 and np.array_equal(st, want["stats"])
        ok &= same
        print(f"{name:44s} FGX_DIRECT={mode} last_direct={how} deferred={ndef} bytes={len(data)} {'OK' if same else 'DIFFERS'}", flush=True)
        if not same:
            if data != want["data"]:
                print("   " + first_diff(data, want["data"]))
            if not np.array_equal(st, want["stats"]):
                print("   counters", st.tolist(), "want", want["stats"].tolist())
    return ok


CASES = {
    "depth 8": (dict(family_size=8), {}, {}, None),
    "depth 8, 3 % errors": (dict(family_size=8, error_rate_ppm=30000), {}, {}, None),
    "depth 5..12 mixed": (dict(family_size=5, family_size_max=12, error_rate_ppm=20000), {}, {}, None),
    "depth 8, min_reads 3": (dict(family_size=8), dict(min_reads=3), dict(min_reads=3), None),
    "depth 4..9, min_reads 6 (orphans)": (dict(family_size=4, family_size_max=9), dict(min_reads=6), dict(min_reads=6), None),
    "depth 8, no per-base tags": (dict(family_size=8), dict(produce_per_base_tags=False), dict(produce_per_base_tags=0), None),
    "depth 6, read-through 151 / insert 120": (dict(family_size=6, read_length=151, insert_mean=120, insert_sd=30), {}, {}, None),
    "depth 8, min input q 30": (dict(family_size=8), dict(min_input_base_quality=30), dict(min_input_base_quality=30), None),
    "depth 8, min input q 38 (strips tails)": (dict(family_size=8), dict(min_input_base_quality=38), dict(min_input_base_quality=38), None),
    "depth 8, no overlap correction": (dict(family_size=8), {}, dict(overlapping_consensus=0), None),
    "depth 8, 250 bp reads": (dict(family_size=8, read_length=250, insert_mean=400), {}, {}, None),
    "depth 8, no cell tag": (dict(family_size=8), dict(cell_tag=None), dict(cell_tag=b"\0\0"), None),
    "long tail 2..50, split forced (merge)": (dict(family_size=2, family_size_max=50), {}, {}, {"FGX_SPLIT": "2"}),
    "depth 8, forced 8 chunks": (dict(family_size=8), {}, {}, {"FGX_SPLIT_CHUNKS": "8"}),
}


def main():
    import subprocess
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=3000)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    ok = True
    for name, (sim, vo, okw, env) in CASES.items():
        if a.only is not None and name != a.only:
            continue
        if env and a.only is None:       # switches that are read once per process: a child with them in its environment
            e = dict(os.environ)
            e.update(env)
            ok &= subprocess.run([sys.executable, os.path.abspath(__file__), "--families", str(a.families), "--only", name], env=e).returncode == 0
            continue
        g = simulate_grouped_reads(a.families, **sim)
        ok &= run(name, g, dict(dict(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), **vo), dict(dict(min_reads=1), **okw))
    if a.only is None:
        print("ALL OK" if ok else "SOME CASES DIFFER")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
