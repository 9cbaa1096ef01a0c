import sys, time, json
sys.path.insert(0, '/root/repo')
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
g = simulate_grouped_reads(n, family_size=8)
c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
out = c.process_batch(g)
t0 = time.perf_counter(); out = c.process_batch(g); dt = time.perf_counter() - t0
print(json.dumps(dict(families=n, raw_reads=g.n_rec, in_bytes=int(g.blob.size), out_bytes=len(out.data), s=dt, reads_per_s=g.n_rec / dt, timing=c.last_timing)))
