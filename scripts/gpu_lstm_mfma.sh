cd "${GRAFT_REPO_ROOT:-/root/repo}"; R="${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4mfma; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_lstm.py -q -s -p no:cacheprovider -k "tolerance or golden_2k" 2>&1 | tail -6 | tee $O/test.txt
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -i "mfma" | head -30 > $R/$O/counters.txt; C=SQ_INSTS_VALU_MFMA_F32; grep -q "SQ_INSTS_VALU_MFMA_MOPS_F32" $R/$O/counters.txt && C="SQ_INSTS_VALU_MFMA_MOPS_F32"; grep -qw "SQ_INSTS_VALU_MFMA_F32" $R/$O/counters.txt && C="SQ_INSTS_VALU_MFMA_F32"; echo "counter $C" > $R/$O/pmc.txt; timeout -k 5 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc -o mfma -- python $R/scripts/gpu_lstm_mfma.py >> $R/$O/pmc.txt 2>&1; echo rc=$? >> $R/$O/pmc.txt )
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob('gpurun_out/r4mfma/pmc/**/*counter_collection*.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','').split('(')[0][:40]; agg[(k,r['Counter_Name'])]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
with open('gpurun_out/r4mfma/mfma_counters.txt','w') as o:
    for (k,c),v in sorted(agg.items()):
        if k.startswith('cmx_'): o.write("%-40s %-32s sum %.0f over %d launches\n" % (k,c,v,n[(k,c)]))
print(open('gpurun_out/r4mfma/mfma_counters.txt').read())
PY
head -12 $O/counters.txt; tail -4 $O/pmc.txt
