set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/g1
python -m pytest tests -m gpu -x -q -k "not cfg4_full" > gpurun_out/g1/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/g1/pytest.log
./tools/bin/ubench_gather 16 > gpurun_out/g1/gather.log 2>&1; cat gpurun_out/g1/gather.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d gpurun_out/g1/gather_pmc -o run -- ./tools/bin/ubench_gather 16 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob('gpurun_out/g1/gather_pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE':
            a=agg[r['Kernel_Name'][:70]]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,v in agg.items(): print(k, v[0], 'launches avg KB', v[1]/v[0])
PY
rm -rf gpurun_out/g1/gather_pmc
python tools/tune_tiers.py cfg4 'kfilter_bits=0' 'kfilter_bits=30' 'kfilter_bits=31' 'kfilter_bits=29' 'kfilter_bits=28' 'kfilter_bits=32' > gpurun_out/g1/tune.log 2>&1; cat gpurun_out/g1/tune.log
python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/g1/bench_cfg4.json 2> gpurun_out/g1/bench_cfg4.err; cat gpurun_out/g1/bench_cfg4.json; tail -3 gpurun_out/g1/bench_cfg4.err
