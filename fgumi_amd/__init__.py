"""fgumi_amd — MI355X-native consensus engine (drop-in for fgumi's per-UMI-family consensus hot path).

The product is the HIP shared library `libfgumi_amd.so` behind the C ABI in `include/fgumi_amd.h`;
this package is the thin Python host mirror of the reference's caller interface
(crates/fgumi-consensus/src/caller.rs:220-252) used by tests, bench.py and integrators.
"""
from ._lib import lib, load, Options, Output, SimParams, default_options, LibraryMissing  # noqa: F401
from .caller import (ConsensusCaller, VanillaUmiConsensusCaller, DuplexConsensusCaller, CodecConsensusCaller, CodecConsensusOptions,  # noqa: F401
                     CodecConsensusStats, VanillaUmiConsensusOptions, ConsensusOutput,
                     ConsensusCallingStats, RejectionReason, GroupedReads, DeviceGroupedReads, DeviceOutput,
                     simulate_grouped_reads, simulated_family_bytes, split_records, MethylationMode)
from .filter import ConsensusFilter, FilterConfig, FilterThresholds, FilterResult, DeviceFilterResult, record_offsets  # noqa: F401,E402
from . import bgzf  # noqa: F401,E402
