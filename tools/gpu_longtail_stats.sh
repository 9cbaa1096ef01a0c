cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for e in "-" "FGX_SPLIT=0"; do if [ "$e" = "-" ]; then E=""; else E="$e"; fi
env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lt -o lt -- python $R/bench.py --families 1000000 --depth 2 --depth-max 50 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "== $e"; python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/lt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        if "sim_generate" in r["Name"]: continue
        print("   %-62s calls %4s avg %9.1f us total/step %8.2f ms"%(r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/4))
PY
rm -rf $R/gpurun_out/lt; done
