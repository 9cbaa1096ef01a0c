"""Stage times of the methylation-aware mode on one GPU: a seeded EM-Seq-like batch (tests/methsim.py) through `fgx_process_batch` with
and without the mode; prints one JSON line (host preparation / kernels / record assembly in ms as fgx_output reports them, reads per
second of the whole call).  The mode runs on the general path (HISTORY.md §13): this is its cost, not a headline."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fgx_opts          # noqa: E402
import methsim           # noqa: E402
from fgumi_amd import GroupedReads          # noqa: E402
from fgumi_amd._lib import Options, Output, lib          # noqa: E402


def run(o, contigs, g, reps=3):
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        if contigs:
            bufs = [C.create_string_buffer(bytes(s), len(s)) for s in contigs]
            ptrs = (C.c_void_p * len(bufs))(*[C.cast(b, C.c_void_p).value for b in bufs])
            lens = (C.c_uint64 * len(bufs))(*[len(s) for s in contigs])
            assert lib.fgx_set_reference(h, len(bufs), ptrs, lens) == 0
        out = Output()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
            dt = time.perf_counter() - t0
            assert rc == 0, lib.fgx_last_error(h).decode()
            if best is None or dt < best["seconds"]:
                best = dict(seconds=round(dt, 4), ms_host_prep=round(out.ms_host_prep, 2), ms_kernels=round(out.ms_kernels, 3), ms_h2d_d2h=round(out.ms_h2d, 2),
                            ms_record_assembly=round(out.ms_emit, 2), consensus_records=int(out.count), reads_per_s=round(g.n_rec / dt))
        return best
    finally:
        lib.fgx_destroy(h)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    rng = methsim.seeded(1234)
    contigs = methsim.genome(rng, n_contigs=4, length=20000)
    g = GroupedReads.from_groups(methsim.simplex_groups(rng, contigs, n, depth=(3, 9), read_len=(80, 150)))
    line = dict(workload=f"{n} EM-Seq-like simplex families ({g.n_rec} reads), host buffers in and out (fgx_process_batch)", host_threads=os.environ.get("FGX_HOST_THREADS", "auto"),
                em_seq=run(fgx_opts.defaults(min_reads=1, methylation_mode=1), contigs, g),
                general_path_without_the_mode=run(fgx_opts.defaults(min_reads=1, track_rejects=1), None, g),
                device_pipeline_without_the_mode=run(fgx_opts.defaults(min_reads=1), None, g))
    print(json.dumps(line))


if __name__ == "__main__":
    main()
