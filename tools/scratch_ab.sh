#!/bin/bash
mkdir -p gpurun_out/r03y
run() { out=$1; shift; timeout -k 10 250 python tools/ab.py "$@" > gpurun_out/r03y/$out.jsonl 2> gpurun_out/r03y/$out.err; python3 -c "
import json,sys
for l in open('gpurun_out/r03y/$out.jsonl'):
    d=json.loads(l); print('$out', d['variant'], d['pass_ms_median'], d['pass_ms_min'], d['pass_ms_max'], d['TBps_of_bytes_read'], d['team_waits']['waits'], d['kernel'][12:100], d['last_solver']['objective'])
"; tail -2 gpurun_out/r03y/$out.err | grep -v amdgpu.ids; }
run fat c5 base "TEAM_MEMBERS=2" "TEAM_MEMBERS=2,TEAM_SLOTS=72" --rounds 5 --steps 60
true
