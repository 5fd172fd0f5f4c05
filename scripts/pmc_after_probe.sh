export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
b() { python $R/bench.py --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 --traffic static 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), 'kernel_ms', round(d['kernel_ms'],3))"; }
b "cold start      "
b "again           "
cd /tmp; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pp1 -o t -- python $R/scripts/prof_frame.py --mode fast --frames 3 > /dev/null 2>&1; cd $R
b "right after PMC "
b "second after PMC"
b "third after PMC "
cd /tmp; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pp2 -o t -- python $R/scripts/prof_frame.py --mode fast --frames 3 > /dev/null 2>&1; cd $R
sleep 8
b "PMC + 8 s sleep "
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp3 -o t -- python $R/scripts/prof_frame.py --mode fast --frames 3 > /dev/null 2>&1; cd $R
b "after --stats   "
