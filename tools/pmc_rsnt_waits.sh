cd /tmp; export TMPDIR=/tmp
R=/root/repo
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_VMEM"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 -f csv --pmc $set --kernel-trace -d $R/gpurun_out/pmc_rsnt/$n -o p -- python $R/tools/bench_rsnt.py --only hann > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/root/repo/gpurun_out/pmc_rsnt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'resonateMfmaBf16' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()): print(k, round(v / n))
PY
