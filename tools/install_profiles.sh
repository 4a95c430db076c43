#!/bin/bash
# Copies what tools/collect_profiles.sh <tag> brought back (gpurun_out/prof_<tag>/) into profiles/<tag>_* and rebuilds
# profiles/pmc_traffic.json (bench.py's roofline.traffic) and the SQ summary quoted in DESIGN.md section 5.
# usage: tools/install_profiles.sh <tag>
set -e
cd "$(dirname "$0")/.."
T=$1; O=gpurun_out/prof_$T; P=profiles
cp $O/stats_f32/t_kernel_stats.csv   $P/${T}_bench_f32_kernel_stats.csv
cp $O/stats_c2f16/t_kernel_stats.csv $P/${T}_bench_config2_f16_kernel_stats.csv
cp $O/stats_c3f16/t_kernel_stats.csv $P/${T}_bench_config3_f16_kernel_stats.csv
cp $O/stats_c3f32/t_kernel_stats.csv $P/${T}_bench_config3_f32_kernel_stats.csv
D=$(cat $O/lib_digest.txt)
for n in stats_f32:bench_f32_under_rocprof stats_c2f16:bench_config2_f16_under_rocprof stats_c3f16:bench_config3_f16_under_rocprof stats_c3f32:bench_config3_f32_under_rocprof; do
    grep -h '^{"metric"' $O/${n%%:*}.log > $P/${T}_${n##*:}.json   # the bench line of the SAME run (HIP-event durations to compare)
done
cp $O/fetch_f32/t_counter_collection.csv $P/${T}_pmc_fetch_size.csv
cp $O/write_f32/t_counter_collection.csv $P/${T}_pmc_write_size.csv
cp $O/fetch_f16/t_counter_collection.csv $P/${T}_pmc_fetch_size_f16.csv
cp $O/write_f16/t_counter_collection.csv $P/${T}_pmc_write_size_f16.csv
cp $O/fetch_c3f32/t_counter_collection.csv $P/${T}_pmc_fetch_size_config3_f32.csv
cp $O/write_c3f32/t_counter_collection.csv $P/${T}_pmc_write_size_config3_f32.csv
# the SQ passes are a row per dispatch and counter (0.5 MB): keep the library's kernels only
for d in f32 f16 c3f32; do
    python3 - $O/sq_$d/t_counter_collection.csv $P/${T}_pmc_sq_$d.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1], newline="")))
k = rows[0].index("Kernel_Name")
w = csv.writer(open(sys.argv[2], "w", newline=""), quoting=csv.QUOTE_NONNUMERIC)
w.writerow(rows[0])
for r in rows[1:]:
    if "mi355" in r[k]:
        w.writerow(r)
PY
done
python3 tools/pmc_aggregate.py --section config2_f32 --fetch $P/${T}_pmc_fetch_size.csv --write $P/${T}_pmc_write_size.csv \
    --section config3_f16 --fetch $P/${T}_pmc_fetch_size_f16.csv --write $P/${T}_pmc_write_size_f16.csv \
    --section config3_f32 --fetch $P/${T}_pmc_fetch_size_config3_f32.csv --write $P/${T}_pmc_write_size_config3_f32.csv \
    --out $P/pmc_traffic.json --digest "$D" \
    --note "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of \`python3 bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline\` (round ${T#r}: one section per workload - config 2 f32, config 3 f16, config 3 f32; tools/collect_profiles.sh); aggregated by tools/pmc_aggregate.py: per-launch means over all launches of the kernel; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section; it also counts MALL hits, so this is an upper bound on DRAM traffic)"
python3 tools/sq_summary.py $P/${T}_pmc_sq_f32.csv $P/${T}_pmc_sq_f16.csv $P/${T}_pmc_sq_c3f32.csv > $P/${T}_pmc_sq_summary.txt
cat $P/${T}_pmc_sq_summary.txt
python3 tools/kernel_clocks.py config3_f16=$P/${T}_pmc_sq_f16.csv config2_f32=$P/${T}_pmc_sq_f32.csv config3_f32=$P/${T}_pmc_sq_c3f32.csv \
    --json $P/kernel_clocks.json --digest "$D" > $P/${T}_kernel_clocks.txt
cat $P/${T}_kernel_clocks.txt
