# Hardware counters of the team kernel next to the streaming GEMV of the set-up (same process, same matrix): separate --pmc
# passes, a few counters each (tools/summarize is not used: the CSVs are small)
set -e
P=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $P/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>> $P/err.log || echo "pass $i failed: $C"
  echo "pass $i: $C"
done
python3 - <<'PY'
import csv, glob, os, collections
P=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02j"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(P+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="k_team_rows" if "k_team_rows" in n else ("gemv_t_kernel" if "gemv_t_kernel" in n else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out=open(P+"/pmc_diag.md","w")
names=sorted({c for k in acc for c in acc[k]})
out.write("| counter (median per launch) | gemv_t_kernel (40 GB streamed, 5.6 ms) | k_team_rows (40 GB, 6.65 ms) |\n|---|---|---|\n")
for c in names:
    med=lambda v: sorted(v)[len(v)//2] if v else float('nan')
    out.write(f"| {c} | {med(acc['gemv_t_kernel'].get(c,[])):.4g} | {med(acc['k_team_rows'].get(c,[])):.4g} |\n")
out.close()
print(open(P+"/pmc_diag.md").read())
PY
rm -rf $P/p[0-9]*
