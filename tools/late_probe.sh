# kernel trace of the three render paths (GPU box): tools/late_probe.sh
cd /tmp && export TMPDIR=/tmp
for m in state mapped image; do
rocprofv3 -f csv --kernel-trace --stats -d /tmp/ws_$m -o t -- python $GRAFT_REPO_ROOT/tools/with_state_probe.py $m > /dev/null 2>&1; f=$(find /tmp/ws_$m -name "*kernel_stats.csv" | head -1); echo "== $m"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
" | head -5; done
