cd $GRAFT_REPO_ROOT
for L in default 16; do
  if [ $L = default ]; then python tools/bench_inflate.py --families 150000 --reps 3; else FGX_INFL_LANES=$L python tools/bench_inflate.py --families 150000 --reps 3; fi
done
python tools/bench_inflate.py --families 60000 --reps 3 --zlib 1
