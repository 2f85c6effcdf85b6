cd /root/repo
timeout 600 python -m pytest tests -m gpu -q -x -k "headline or config5" 2>&1 | tail -2
for i in 1 2 3; do
for tag in dum0 hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_loop'])"
done; done
unset DEPTHG_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r04/pmc_dum -- python3 /root/repo/bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/root/repo/gpurun_out/r04/pmc_dum/*/*counter_collection.csv")[0]
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]; acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    if "k_corr2" in k or "k_gs" in k: print(k, "FETCH MB", acc[k]/len(n[k])*1024*2/1e6)
PY
