#!/bin/bash
# copies what tools/collect_profiles.sh left in gpurun_out/prof_<tag> into profiles/ and regenerates the derived tables
TAG=${1:-r04}; O=gpurun_out/prof_$TAG
# refuse to touch profiles/ unless the collection is complete: a failed or unscheduled gpurun call leaves nothing here,
# and the redirects below would otherwise truncate the committed files
set -e
for f in $O/bench_line.json $O/bench_line_profiled.json $O/c1_line.json $O/c1_line_profiled.json $O/c3_line.json $O/c3_line_profiled.json; do
    [ -s "$f" ] || { echo "install_profiles: $f missing or empty; profiles/ left untouched" >&2; exit 1; }
done
for d in bench_stats c1_stats c3_stats pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ_BUSY_CU_CYCLES; do
    ls $O/$d/runc/*.csv >/dev/null 2>&1 || { echo "install_profiles: no csv under $O/$d/runc; profiles/ left untouched" >&2; exit 1; }
done
par=$(ls $O/bench_stats/runc/*_kernel_stats.csv | sort -V | head -1); chi=$(ls $O/bench_stats/runc/*_kernel_stats.csv | sort -V | tail -1)
cp $par profiles/${TAG}_bench_kernel_stats.csv; cp $chi profiles/${TAG}_bench_extras_child_kernel_stats.csv
cp $O/c1_stats/runc/*_kernel_stats.csv profiles/${TAG}_c1_kernel_stats.csv; cp $O/c3_stats/runc/*_kernel_stats.csv profiles/${TAG}_c3_kernel_stats.csv
for w in bench c1 c3; do tail -1 $O/${w}_line.json > profiles/${TAG}_${w}_line.json; tail -1 $O/${w}_line_profiled.json > profiles/${TAG}_${w}_line_profiled.json; done
cp $O/pmc_FETCH_SIZE/runc/*counter_collection.csv profiles/${TAG}_pmc_FETCH_SIZE_fit_n16384.csv
cp $O/pmc_WRITE_SIZE/runc/*counter_collection.csv profiles/${TAG}_pmc_WRITE_SIZE_fit_n16384.csv
cp $O/pmc_SQ_BUSY_CU_CYCLES/runc/*counter_collection.csv profiles/${TAG}_pmc_MFMA_fit_n16384.csv
python tools/summarize_profiles.py $TAG
