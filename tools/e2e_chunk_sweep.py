"""The file -> file leg of bench.py (fgx_run_bam) over chunk sizes: one input file per size, every size twice (best of two inside
bench.end_to_end).  Run on the GPU box:
    [E2E_SWEEP_MB=512,256,128,64,32] python tools/e2e_chunk_sweep.py [families ...]      (default 250000 1000000)"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions

c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
d = "/tmp/fgx_e2e_sweep"
os.makedirs(d, exist_ok=True)
rows = []
for fam in [int(a) for a in sys.argv[1:]] or [250000, 1000000]:
    path = os.path.join(d, f"grouped_{fam}.bam")
    n_rec = bench.write_grouped_bam(path, fam, 8, 150)
    for ahead in [0]:
        for mb in [int(a) for a in os.environ.get("E2E_SWEEP_MB", "512,256,128,64,32").split(",")]:
            r = bench.end_to_end(c, fam, 8, 150, d, chunk_mb=mb, grouped=(path, n_rec))
            row = dict(families=fam, ahead=ahead, chunk_mb=mb, M_reads_s=round(r["value"] / 1e6, 1), chunks=r["chunks"], total_s=round(r["total_s"], 4), bottleneck=r["bottleneck"],
                       busy={k: round(v, 4) for k, v in r["stage_busy_s"].items()}, device={k: round(v, 4) for k, v in r["device_stage_s"].items()})
            rows.append(row)
            print(json.dumps(row), flush=True)
    os.remove(path)
c.close()
