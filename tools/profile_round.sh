#!/bin/bash
# One measurement round on the GPU box (run through gpurun): the driver's bench line, the rocprofv3 kernel statistics
# of the same command and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md
# prescribes), then profiles/summarize.py turns them into the committed summaries.
#   usage: bash tools/profile_round.sh <tag>        (writes gpurun_out/<tag>/ and profiles/<tag>_*)
set -o pipefail
TAG=${1:-r02}
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
echo "summaries written"
