cd /root/repo
for i in 1 2; do
for tag in hip nt64; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  echo "== $tag"
  TAG=$tag scripts/kstats.sh headline+head 2>&1 | grep -E "k_head|ms_per" | cut -c1-250 | sed 's/"host_ms.*//'
done; done
timeout 600 python -m pytest tests/test_gpu_head.py -q -x 2>&1 | tail -2
DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_nt64.so timeout 600 python -m pytest tests/test_gpu_head.py -q -x 2>&1 | tail -2
