out=$(realpath gpurun_out/r2z); root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -- python3 $root/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-all-matched --no-single-gpu-reference > $out/prof.json 2> $out/prof.err
cd $root
db=$(find $out/prof -name "*.db" | head -1)
python3 scripts/profile_summary.py stats $db $out/prof_stats.csv
rm -rf $out/prof
head -30 $out/prof_stats.csv
