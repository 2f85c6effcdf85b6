cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -x -k "total_grad_only" 2>&1 | tail -15
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default ', d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 300 python bench.py --no-cpu-baseline --total-grad-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('total-only', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
timeout 300 python bench.py --no-cpu-baseline --total-grad-only > gpurun_out/r04/bench_tgo.json 2> gpurun_out/r04/bench_tgo.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04/prof_tgo -o tgo -- python3 /root/repo/bench.py --no-cpu-baseline --total-grad-only --steps 50 --warmup 5 > /dev/null 2>&1
cd /root/repo; ls gpurun_out/r04/prof_tgo/ | head; python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r04/prof_tgo/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
