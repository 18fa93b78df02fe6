set -e
# One profiling round of bench.py on the GPU box: the plain line, the rocprofv3 kernel trace (+ stats), and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE — separate runs, no other trace domain: MI355X_MICROARCH.md).   bash tools/prof_round.sh <tag> [bench args]
TAG=${1:-r03}; shift || true
P=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py "$@" > $P/bench_line.json 2> $P/bench_err.log
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" > $P/bench_line_under_rocprof.json 2>> $P/bench_err.log
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>> $P/bench_err.log
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>> $P/bench_err.log
echo "write done"
cd $GRAFT_REPO_ROOT && python3 tools/summarize_prof.py $P $P/summary "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of ${TAG} (profiles/${TAG}_rocprofv3_summary.*; FETCH_SIZE x2: the gfx950 correction of MI355X_MICROARCH.md), same build" > $P/summary_stdout.md

f=$(find $P/trace -name "*kernel_stats.csv" | head -1); cp $f $P/kernel_stats.csv
# keep the merge small: drop the raw traces
rm -rf $P/trace $P/pmc_fetch $P/pmc_write
head -40 $P/summary.md
