#!/bin/bash
# Closes "whole-BAM parity unpinned" on a box that has BOTH a Rust toolchain and an MI355X:
#   1. build the reference with its compare tool          (docs/compare-cli.md)
#   2. export the simulated grouped input + this engine's output as BAM files   (tools/export_bam.py)
#   3. run the reference's own caller on the SAME grouped.bam
#   4. `fgumi compare bams` → exit code 0 = identical consensus bases, quals, tags, record order
# usage: tools/ref_pin.sh /path/to/fgumi-checkout [simplex|duplex|codec] [families] [depth]
set -euo pipefail
REF=${1:?path to a fulcrumgenomics/fgumi checkout}; CALLER=${2:-simplex}; FAM=${3:-200000}; DEPTH=${4:-8}
HERE=$(cd "$(dirname "$0")/.." && pwd); OUT=${OUT:-/tmp/fgx_pin_$CALLER}; mkdir -p "$OUT"
(cd "$REF" && cargo build --release --features simulate,compare)
FGUMI="$REF/target/release/fgumi"
python "$HERE/tools/export_bam.py" --caller "$CALLER" --families "$FAM" --depth "$DEPTH" --out-dir "$OUT"
case "$CALLER" in
  simplex) "$FGUMI" simplex -i "$OUT/grouped.bam" -o "$OUT/ref.bam" --min-reads 1 --threads "$(nproc)" ;;
  duplex)  "$FGUMI" duplex  -i "$OUT/grouped.bam" -o "$OUT/ref.bam" --min-reads 1 --threads "$(nproc)" ;;
  codec)   "$FGUMI" codec   -i "$OUT/grouped.bam" -o "$OUT/ref.bam" --threads "$(nproc)" ;;
esac
"$FGUMI" compare bams "$OUT/ref.bam" "$OUT/ours.bam" --command "$CALLER" && echo "PINNED: $CALLER output identical to the reference build"
