#!/bin/bash
# One measurement round on the GPU box (run through gpurun): the driver's bench line, the rocprofv3 kernel statistics
# of the same command and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md
# prescribes), then profiles/summarize.py turns them into the committed summaries.
# Last step: tools/sweep_diff.py holds the new bench line against the PREVIOUS committed round (second argument, default:
# the newest profiles/*_bench.json that is not this tag's) and the script FAILS LOUDLY (exit 3, after every file has
# been written) when a cell of the 8-32-bit band lost more than 5 % under both normalisations of tools/sweep_diff.py.
#   usage: bash tools/profile_round.sh <tag> [previous_bench.json]   (writes gpurun_out/<tag>/ and profiles/<tag>_*)
set -o pipefail
TAG=${1:-r03}
PREV=${2:-$(ls -t profiles/r*_bench.json 2>/dev/null | grep -v "profiles/${1:-r03}_bench.json" | grep -v configs | head -1)}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
echo "bench done"
SHORT="--steps 20 --warmup 3 --no-sweep --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 bench.py $SHORT > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" || exit 1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-sweep --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-sweep --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.err" || exit 1
echo "pmc done"
python3 profiles/summarize.py "$TAG" "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" 100000000 > "$OUT/summary.json" || exit 1
cp "$OUT/bench.json" "profiles/${TAG}_bench.json"
cp "$OUT/bench_under_rocprof.json" "profiles/${TAG}_bench_under_rocprof.json"
mkdir -p "$OUT/profiles_out" && cp profiles/${TAG}_* profiles/pmc_traffic.json "$OUT/profiles_out/"
cp duckdb-adaptive-compression_amd/build/adac_kernels.resources.txt "$OUT/kernel_resources_remarks.txt" 2>/dev/null
python3 tools/kernel_resources.py > "profiles/${TAG}_kernel_resources.txt" 2>/dev/null && cp "profiles/${TAG}_kernel_resources.txt" "$OUT/profiles_out/"
echo "summaries written"
if [ -n "$PREV" ] && [ -f "$PREV" ]; then
  echo "sweep diff against $PREV"
  python3 tools/sweep_diff.py "$PREV" "$OUT/bench.json" | tee "$OUT/sweep_diff.txt"
  RC=${PIPESTATUS[0]}
  cp "$OUT/sweep_diff.txt" "profiles/${TAG}_sweep_diff_vs_$(basename "$PREV" _bench.json).txt"
  cp profiles/${TAG}_sweep_diff_vs_* "$OUT/profiles_out/"
  if [ "$RC" != "0" ]; then echo "!!!! THROUGHPUT REGRESSION against $PREV (see $OUT/sweep_diff.txt) !!!!"; exit 3; fi
fi
