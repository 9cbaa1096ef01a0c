import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions
c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
for mb in (512, 256, 128, 64, 32):
    os.environ["FGX_BENCH_E2E_CHUNK_MB"] = str(mb)
    r = bench.end_to_end(c, 250000, 8, 150, "/tmp/fgx_e2e_test")
    print(mb, "MB chunks:", round(r["value"] / 1e6, 1), "M reads/s", r["chunks"], "chunks", round(r["total_s"], 4), "s", r["bottleneck"], {k: round(v, 4) for k, v in r["device_stage_s"].items()})
c.close()
