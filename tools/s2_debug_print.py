import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads
g = simulate_grouped_reads(8, family_size=8)
c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True, device=0)
out = c.process_batch(g)
print(out.count)
c.close()
