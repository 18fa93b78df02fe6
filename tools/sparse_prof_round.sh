set -e
# The sparse products under rocprofv3 at two shapes — the 1 M x 100 k matrix of the bench's sparse leg and covtype's 581 012 x 54 —
# kernel trace (+ stats) and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no other trace domain), summarised by
# tools/summarize_prof.py.   bash tools/sparse_prof_round.sh <tag>      
TAG=${1:-r05s}
P=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for shape in "1000000 100000 60 1Mx100k" "581012 54 11 covtype"; do
  set -- $shape
  Q=$P/$4
  mkdir -p $Q
  rocprofv3 --kernel-trace --stats --output-format csv -d $Q/trace -- python3 $GRAFT_REPO_ROOT/tools/sparse_prof.py $1 $2 $3 > $Q/log.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $Q/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/sparse_prof.py $1 $2 $3 >> $Q/log.txt 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $Q/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/sparse_prof.py $1 $2 $3 >> $Q/log.txt 2>&1
  (cd $GRAFT_REPO_ROOT && python3 tools/summarize_prof.py $Q $Q/summary "sparse_logreg on a sparse X $1 x $2, ~$3 entries per row, 50 iterations (tools/sparse_prof.py)" > $Q/summary_stdout.md)
  rm -rf $Q/trace $Q/pmc_fetch $Q/pmc_write
  echo "$4 done"
done
