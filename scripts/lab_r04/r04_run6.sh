cd /root/repo
mkdir -p gpurun_out/r04
for v in hip fixnop fixrot; do
  echo "=== $v vs nocorr2"; CMP_HW=40 python scripts/cmp_ws.py $v nocorr2 2 dense 2>&1 | grep -v "^/opt\|Warn" | grep "^hip\|^fix\|^nocorr2\|group 0\|tile 0 lane\|gradient-tile\|per channel" | cut -c1-700
done > gpurun_out/r04/cmpws_variants.txt 2>&1
cat gpurun_out/r04/cmpws_variants.txt
